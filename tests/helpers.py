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

# ---- Prithvi fixtures (tests/golden/make_golden_prithvi.py) --------------------------------------------
PRITHVI_SMALL = dict(img_size=32, patch_size=8, num_frames=1, tubelet_size=1, in_chans=3, embed_dim=32, depth=2, num_heads=2,
                     decoder_embed_dim=16, decoder_depth=1, decoder_num_heads=2)
PRITHVI_SMALL_T3 = dict(PRITHVI_SMALL, num_frames=3)
PRITHVI_SEG_SMALL = dict(PRITHVI_SMALL, img_size=64, patch_size=16)
PRITHVI_FULL = dict(img_size=224, patch_size=16, num_frames=1, tubelet_size=1, in_chans=6, embed_dim=768, depth=12, num_heads=12,
                    decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16)
MAE_CASES = {
    # tag: (args, B, mask_ratio, seed, grads)
    "small_bs2": (PRITHVI_SMALL, 2, 0.75, 11, True),
    "small_t3_bs2": (PRITHVI_SMALL_T3, 2, 0.75, 12, True),
    "small_r0_bs2": (PRITHVI_SMALL, 2, 0.0, 13, False),
    "full_bs1": (PRITHVI_FULL, 1, 0.75, 14, False),
}
SEG_CASES = {
    # tag: (args, B, ncls, fcn_out, frozen, train, seed)
    "small_eval": (PRITHVI_SEG_SMALL, 2, 4, 8, True, False, 21),
    "small_train_frozen": (PRITHVI_SEG_SMALL, 2, 4, 8, True, True, 22),
    "small_train_unfrozen": (PRITHVI_SEG_SMALL, 2, 4, 8, False, True, 23),
    "full_eval_bs1": (PRITHVI_FULL, 1, 4, 256, True, False, 24),
}


def mae_inputs(tag):
    from oracle import detgen
    from oracle import prithvi_ref as P

    args, B, ratio, seed, grads = MAE_CASES[tag]
    cfg = P.MaeCfg(**args)
    sd = detgen.fill_state(P.mae_state_shapes(cfg), seed=seed)
    sd["pos_embed"] = P.sincos_pos_embed(cfg.embed_dim, cfg.grid)
    sd["decoder_pos_embed"] = P.sincos_pos_embed(cfg.decoder_embed_dim, cfg.grid)
    x = detgen.normal(f"{tag}.x", (B, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), seed=seed)
    noise = detgen.uniform(f"{tag}.noise", (B, cfg.num_patches), 0.0, 1.0, seed=seed)
    return cfg, sd, x, noise, ratio


def seg_inputs(tag):
    from oracle import detgen
    from oracle import prithvi_ref as P

    args, B, ncls, fcn_out, frozen, train, seed = SEG_CASES[tag]
    m = P.MaeCfg(**args)
    cfg = P.SegCfg(mae=m, num_classes=ncls, fcn_out_channels=fcn_out, fcn_num_convs=1, fcn_dropout=0.1, frozen_backbone=frozen)
    sd = detgen.fill_state(P.seg_state_shapes(cfg), seed=seed)
    sd["backbone.pos_embed"] = P.sincos_pos_embed(m.embed_dim, m.grid)
    sd["backbone.decoder_pos_embed"] = P.sincos_pos_embed(m.decoder_embed_dim, m.grid)
    x = detgen.normal(f"{tag}.x", (B, m.in_chans, m.num_frames, m.img_size, m.img_size), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, m.img_size, m.img_size), ncls, seed=seed)
    noise = detgen.uniform(f"{tag}.noise", (B, m.num_patches), 0.0, 1.0, seed=seed)
    drop_u = detgen.uniform(f"{tag}.drop", (B, fcn_out), 0.0, 1.0, seed=seed)
    return cfg, sd, x, y, noise, drop_u, train
