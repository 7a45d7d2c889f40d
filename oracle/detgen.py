"""TEST INFRASTRUCTURE ONLY — deterministic, torch-RNG-independent tensor generator.

Golden fixtures are produced in the build container (where /root/reference is importable) and
consumed on the GPU box (where it is not).  Both sides must materialise bit-identical weights
and inputs without shipping state-dicts, so values come from a counter-based integer hash
(splitmix64) of (seed, fnv1a(name), element index) — pure numpy uint64 arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, shape, seed: int = 0) -> np.ndarray:
    """float64 uniform in [0, 1) with 53 random bits per element."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    bits = _splitmix64(idx) >> np.uint64(11)
    return (bits.astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def uniform(name, shape, lo, hi, seed=0) -> torch.Tensor:
    return torch.from_numpy((lo + (hi - lo) * uniform01(name, shape, seed)).astype(np.float32))


def normal(name, shape, std=1.0, seed=0) -> torch.Tensor:
    """Box-Muller on two hashed uniforms."""
    u1 = uniform01(name + "#1", shape, seed)
    u2 = uniform01(name + "#2", shape, seed)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((std * z).astype(np.float32))


def labels(name, shape, num_classes: int, p_zero: float = 0.05, seed=0) -> torch.Tensor:
    """int64 labels: class 0 with prob p_zero (the ignored class), else uniform over 1..C-1."""
    u = uniform01(name, shape, seed)
    v = uniform01(name + "#c", shape, seed)
    y = 1 + np.floor(v * (num_classes - 1)).astype(np.int64)
    y = np.where(u < p_zero, 0, np.minimum(y, num_classes - 1))
    return torch.from_numpy(y.astype(np.int64))


def fill_state(shapes: dict[str, tuple], seed: int = 0, gain: float = 1.0) -> dict[str, torch.Tensor]:
    """Weights for a reference-named state dict: conv/linear ~ U(+-sqrt(3/fan_in))*gain so
    activations stay O(1); BN gamma ~ U(.5,1.5), beta ~ U(-.2,.2), running stats non-trivial."""
    sd: dict[str, torch.Tensor] = {}
    for name, shp in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(3, dtype=torch.int64)
        elif name.endswith("running_mean"):
            sd[name] = uniform(name, shp, -0.1, 0.1, seed)
        elif name.endswith("running_var"):
            sd[name] = uniform(name, shp, 0.5, 1.5, seed)
        elif len(shp) == 1 and (".ln." in name or "norm" in name or _is_bn_affine(name, shapes)):
            if name.endswith("weight"):
                sd[name] = uniform(name, shp, 0.5, 1.5, seed)
            else:
                sd[name] = uniform(name, shp, -0.2, 0.2, seed)
        elif len(shp) == 1:  # conv / linear bias
            sd[name] = uniform(name, shp, -0.1, 0.1, seed)
        elif name.endswith("pos_embed"):
            sd[name] = torch.zeros(shp)  # overwritten by the sincos table
        elif name in ("cls_token", "mask_token") or name.endswith(".cls_token") or name.endswith(".mask_token"):
            sd[name] = normal(name, shp, 0.02, seed)
        else:
            fan_in = int(np.prod(shp[1:]))
            if name.startswith("up_convs") or name.startswith("input_up_conv") or "feature_pyramid_net" in name:
                fan_in = shp[0]  # ConvTranspose2d weight is [Cin, Cout, 2, 2]; each output sees Cin taps
            a = gain * (3.0 / max(fan_in, 1)) ** 0.5
            sd[name] = uniform(name, shp, -a, a, seed)
    return sd


def _is_bn_affine(name: str, shapes) -> bool:
    stem = name.rsplit(".", 1)[0]
    return (stem + ".running_mean") in shapes
