"""Which stages of a bf16-MIXED plan carry FLAG_BF16: the shape lists of csrc/conv_bf16.hip (launch_conv_bf16) and
csrc/wgrad_bf16.hip (launch_wgrad_bf16), restated for the planner.  A flagged stage rounds its two MFMA operands to bf16 (f32
accumulate); everything else of the plan computes in exact f32.  For a flagged stage whose operands are all f32 the native launchers
fall back to the f32 kernels when they decline a shape (speed, not correctness).  That is NOT so once an operand is STORED as bf16
(CONV.X1_BF16 / WGRAD.P_BF16, plan/unet_plan._dy_bf16_ok): only the bf16 kernels can read such a tensor, so a stage the native
launcher declines - its shape list, but also its internal guards (the 2 GiB buffer-range check of wgrad_bf16.hip, the LDS / shape
fall-backs of conv_bf16.hip's launch_b16) - FAILS the step with S2K_EINVAL rather than computing on garbage.  The planner
therefore only stores dY as bf16 where these lists accept every reader; tests/test_bf16_mixed_gpu.py checks on real plans that
every flagged stage ran on the bf16 kernels (s2k_program_profile_variants), and the method-path / encoder-only planners
(plan_mae_encoder / _decoder / _loss, plan_random_masking, encoder_plan) do not implement the mode: they plan exact f32 whatever
`precision` says (the fused forward is the path the mode is for)."""
from __future__ import annotations

from . import opdefs as D

_PIX_PRO = (D.PRO_NONE, D.PRO_AFFINE, D.PRO_SILU, D.PRO_RELU)
_W3 = (16, 32, 56, 28, 14, 112, 224)


def conv_ok(f: dict) -> bool:
    """CONV record `f` (planner fields) is one of conv_bf16.hip's shapes."""
    if f["STRIDE"] != 1 or f["HO"] != f["H"] or f["WO"] != f["W"]:
        return False
    T = f["KH"] * f["KW"]
    hw = f["H"] * f["W"]
    gate = f.get("GATE1") is not None
    if f["MODE"] == D.MODE_CONVT_SCATTER:
        return T == 1 and f["C2"] == 0 and hw % 4 == 0 and not gate and f["M"] > 32 and f["M"] % 4 == 0 and f["PRO1"] in (D.PRO_RELU, D.PRO_SILU)
    if f["MODE"] != D.MODE_CONV:
        return False
    if T == 1:
        if f["C2"] != 0 or hw % 4 or f["PAD_T"] or f["PAD_L"]:
            return False
        return f["PRO1"] == D.PRO_SILU if gate else f["PRO1"] in _PIX_PRO
    if T != 9 or f["KH"] != 3 or f["PAD_T"] != 1 or f["PAD_L"] != 1 or gate or f["PRO1"] not in (D.PRO_NONE, D.PRO_RELU):
        return False
    if f["C2"] > 0 and (f["PRO2"] != f["PRO1"] or f["C1"] % 16):
        return False
    if f["M"] <= 32:
        return f["WO"] >= 64 and f["WO"] % 64 == 0
    return (f["WO"] >= 64 and f["WO"] % 64 == 0) or f["WO"] in _W3


def wgrad_ok(f: dict) -> bool:
    """WGRAD record `f` is one of wgrad_bf16.hip's shapes."""
    if f["MODE"] != D.MODE_CONV or f["STRIDE"] != 1 or f["H"] != f["HO"] or f["W"] != f["WO"] or f.get("GATEP") is not None:
        return False
    T = f["KH"] * f["KW"]
    hw = f["HO"] * f["WO"]
    if T == 1:
        return hw % 8 == 0 and f["B"] * hw >= 512
    if T != 9 or f["KH"] != 3 or f["PAD_T"] != 1 or f["PAD_L"] != 1 or f.get("GATEQ") is not None or f["PROP"] != D.PRO_NONE:
        return False
    return f["WO"] % 64 == 0 or f["WO"] % 56 == 0 or f["WO"] in (32, 16)
