"""Shared helpers for the parity tests (fixture loading, subsampling identical to make_golden)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def sub(t: torch.Tensor, n: int = 2048) -> np.ndarray:
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).cpu().numpy().copy()


def checks(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double().cpu()
    return np.array([d.sum().item(), d.abs().sum().item(), d.abs().max().item(), float(d.numel())])


def load(name: str):
    return np.load(GOLDEN / name, allow_pickle=False)


def rel_err(a, b) -> float:
    """max |a-b| / max|b| — the '1e-3 relative' bar of BASELINE.json north_star is applied to this."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


UNET_CASES = {
    # tag: (version, C, H, B, ncls, train, seed)
    "b0_224_eval_bs1": ("b0", 6, 224, 1, 4, False, 1),
    "b0_224_eval_bs2": ("b0", 6, 224, 2, 4, False, 2),
    "b0_224_train_bs2": ("b0", 6, 224, 2, 4, True, 3),
    "b5_224_eval_bs1": ("b5", 6, 224, 1, 4, False, 4),
    "b0_128x4_eval_bs1": ("b0", 4, 128, 1, 4, False, 5),
    "b0_128x4_train_bs2": ("b0", 4, 128, 2, 4, True, 6),
    "b5_256x13_eval_bs1": ("b5", 13, 256, 1, 4, False, 7),
    "b5_64x13_train_bs2": ("b5", 13, 64, 2, 4, True, 8),
}
