"""Run planned s2k programs through the CPU emulator (oracle/ops_ref.py) — test infrastructure."""
from __future__ import annotations

import numpy as np
import torch

import s2lc_amd  # noqa: F401
from s2lc_amd.plan import opdefs as D
from oracle import ops_ref

NARROW = (D.BASE["CONST"],)


def _bytes(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().reshape(-1).view(torch.uint8)


def make_bases(plan, flat_params, flat_bufs, x, noise, n_out, wide: bool = False):
    """wide: hold every f32 tensor as float64 (see ops_ref.Mem)."""
    fd = torch.float64 if wide else torch.float32
    k = 2 if wide else 1
    pad8 = lambda n: (n + 7) // 8 * 8  # noqa: E731
    bases = {
        D.BASE["WS"]: torch.zeros(k * pad8(plan.ws_bytes) + 64, dtype=torch.uint8),
        D.BASE["AUX"]: torch.zeros(k * pad8(plan.aux_bytes) + 64, dtype=torch.uint8),
        D.BASE["PARAMS"]: _bytes(flat_params.detach().to(fd).clone()),
        D.BASE["GRADS"]: _bytes(torch.zeros_like(flat_params, dtype=fd)),
        D.BASE["WGS"]: _bytes(torch.zeros_like(flat_params, dtype=fd)),
        D.BASE["BUFS"]: _bytes(flat_bufs.detach().to(fd).clone()),
        D.BASE["X"]: _bytes(x.to(fd).clone()),
        D.BASE["OUT"]: _bytes(torch.zeros(n_out, dtype=fd)),
        D.BASE["DOUT"]: _bytes(torch.zeros(n_out, dtype=fd)),
        D.BASE["NOISE"]: _bytes(noise.clone().to(fd)),
        D.BASE["CONST"]: _bytes(torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32)),
        D.BASE["WPACK"]: torch.zeros(k * pad8(plan.wpack_bytes) + 64, dtype=torch.uint8),
        D.BASE["DX"]: _bytes(torch.full(tuple(x.shape), float("nan"), dtype=fd)),
    }
    # poison the workspace so that reads of never-written memory show up as NaN
    bases[D.BASE["WS"]][: k * pad8(plan.ws_bytes)].view(fd).fill_(float("nan"))
    return bases


def fview(bases, name: str, wide: bool = False) -> torch.Tensor:
    return bases[D.BASE[name]].view(torch.float64 if wide else torch.float32)


def emulate(packed: np.ndarray, bases, wide: bool = False) -> None:
    ops_ref.run_program(packed, bases, D, wide=wide, narrow_bases=NARROW)


def make_bases_vit(plan, flat_params, flat_bufs, x, noise_f32: torch.Tensor, wide: bool = False):
    """Bases for a VitPlan (plan/vit_plan.py): OUT / DOUT / NOISE are sized from the plan."""
    fd = torch.float64 if wide else torch.float32
    k = 2 if wide else 1
    pad8 = lambda n: (n + 7) // 8 * 8  # noqa: E731
    n_dout = int(np.prod(plan.dout_shape))
    nz = torch.zeros(plan.noise_bytes // 4, dtype=fd)
    nz[: noise_f32.numel()] = noise_f32.reshape(-1).to(fd)
    bases = {
        D.BASE["WS"]: torch.zeros(k * pad8(plan.ws_bytes) + 64, dtype=torch.uint8),
        D.BASE["AUX"]: torch.zeros(k * pad8(plan.aux_bytes) + 64, dtype=torch.uint8),
        D.BASE["PARAMS"]: _bytes(flat_params.detach().to(fd).clone()),
        D.BASE["GRADS"]: _bytes(torch.zeros_like(flat_params, dtype=fd)),
        D.BASE["WGS"]: _bytes(torch.zeros_like(flat_params, dtype=fd)),
        D.BASE["BUFS"]: _bytes(flat_bufs.detach().to(fd).clone()),
        D.BASE["X"]: _bytes(x.to(fd).clone()),
        D.BASE["OUT"]: torch.zeros(k * pad8(plan.out_bytes) + 64, dtype=torch.uint8),
        D.BASE["DOUT"]: (torch.zeros(k * pad8(plan.dout_bytes) + 64, dtype=torch.uint8) if getattr(plan, "dout_bytes", 0)
                         else _bytes(torch.zeros(n_dout, dtype=fd))),
        D.BASE["DX"]: torch.zeros(k * pad8(x.numel() * 4) + 64, dtype=torch.uint8),
        D.BASE["NOISE"]: _bytes(nz),
        D.BASE["CONST"]: _bytes(torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32)),
        D.BASE["WPACK"]: torch.zeros(k * pad8(plan.wpack_bytes) + 64, dtype=torch.uint8),
    }
    bases[D.BASE["WS"]][: k * pad8(plan.ws_bytes)].view(fd).fill_(float("nan"))
    return bases


def flat_from_state(layout, sd, dtype=torch.float32):
    """Flat parameter / buffer vectors in the plan's layout from a reference-named state dict."""
    fp = torch.zeros(layout.n_params, dtype=dtype)
    fb = torch.zeros(max(layout.n_bufs, 1), dtype=dtype)
    for name, (off, shape) in layout.params.items():
        fp[off:off + int(np.prod(shape))] = sd[name].detach().reshape(-1).to(dtype)
    for name, (off, shape) in layout.bufs.items():
        fb[off:off + int(np.prod(shape))] = sd[name].detach().reshape(-1).to(dtype)
    return fp, fb


def out_view(bases, tref, wide: bool = False) -> torch.Tensor:
    """A tensor of the OUT base (any dtype) of a VitPlan."""
    mem = ops_ref.Mem(bases, wide, NARROW)
    return mem.view(tref.ref, tref.shape, tref.dtype)
