"""GPU parity, stage by stage: every HIP kernel family is launched through the C ABI
(`s2k_program_run` on a one-record program) and compared with the oracle's CPU statement of the
same stage (oracle/ops_ref.py) on identical seeded bytes.  fp32 tolerance is stated per case
(float atomics make sums order-dependent; 1e-4 relative to the tensor's max is the bar here, the
network-level bar of BASELINE.json is 1e-3)."""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import ops_ref
from s2lc_amd.plan import opdefs as D
from s2lc_amd.plan.program import Arena, Program

pytestmark = pytest.mark.gpu

WS = D.BASE["WS"]
_DT = {"f32": torch.float32, "f64": torch.float64, "i64": torch.int64, "i32": torch.int32, "i16": torch.int16, "u8": torch.uint8,
       "bf16": torch.bfloat16}


class Case:
    def __init__(self, seed=0):
        self.arena = Arena(WS)
        self.items = {}
        self.gen = torch.Generator().manual_seed(seed)

    def t(self, name, shape, fill="randn", dtype="f32", scale=1.0):
        ref = self.arena.alloc(name, shape, dtype)
        if isinstance(fill, torch.Tensor):
            data = fill.to(_DT[dtype]).reshape(shape).clone()
        elif fill == "randn":
            data = torch.randn(shape, generator=self.gen) * scale
        elif fill == "rand":
            data = torch.rand(shape, generator=self.gen) * scale
        elif fill == "pos":
            data = torch.rand(shape, generator=self.gen) * scale + 0.5
        elif fill == "zeros":
            data = torch.zeros(shape)
        elif fill == "nan":
            data = torch.full(shape, float("nan"))
        else:
            raise ValueError(fill)
        self.items[name] = (ref, data.to(_DT[dtype]))
        return ref

    def bnv(self, name, C):
        """{scale, shift, mean, invstd} of a plausible BatchNorm."""
        scale = torch.rand(C, generator=self.gen) + 0.5
        shift = torch.randn(C, generator=self.gen) * 0.3
        mean = torch.randn(C, generator=self.gen) * 0.3
        invstd = torch.rand(C, generator=self.gen) + 0.7
        return self.t(name, (4, C), torch.stack([scale, shift, mean, invstd]))

    def pack(self, wt, M, K, T, s_m, s_k, s_t, flip, src_elem_off=0, bf16=False, q4=False):
        """WEIGHT_PACK record for one weight tensor living in this arena; returns (pre-op, packed ref, MP) and, with bf16, the
        reference of the bf16 copy as a fourth element."""
        MP, KP = (M + 127) // 128 * 128, (K + 63) // 64 * 64
        dst = self.t(f"packed{len(self.items)}", (KP * T, MP), "nan")
        row = [wt.off // 4 + src_elem_off, dst.off // 4, M, K, T, s_m, s_k, s_t, flip | (2 if q4 else 0), MP, KP, 0]
        tab = self.t(f"packtab{len(self.items)}", (1, 12), torch.tensor([row]), "i32")
        zero = self.arena.alloc("zero", (1,))  # offsets in the table are relative to SRC / DST = arena start
        fields = dict(TABLE=tab, SRC=zero.at(-(zero.off // 4)), DST=zero.at(-(zero.off // 4)), TOTAL=KP * T * MP, N_ENTRIES=1)
        if q4:      # the f32 quad copy of a 1x1 entry lives at DST bytes + Q4_BASE + 4 * dst_off
            dstq = self.t(f"packedq_{len(self.items)}", (KP * T, MP), "nan")
            fields["Q4_BASE"] = dstq.off - dst.off
            return ("WEIGHT_PACK", fields), dst, MP, dstq
        if not bf16:
            return ("WEIGHT_PACK", fields), dst, MP
        # the bf16 copy of an entry lives at DST bytes + BF16_BASE + 2 * dst_off (plan: a mirror region behind the f32 packs)
        dst16 = self.t(f"packed16_{len(self.items)}", (KP * T * MP // 2,), "nan")
        fields["BF16_BASE"] = dst16.off - dst.off // 2
        return ("WEIGHT_PACK", fields), dst, MP, dst16

    def run(self, kind, outputs, tol=1e-4, sum0=(), pre=(), want_variant=None, **fields):
        """sum0: outputs compared after summing their leading (statistics-replica) dimension.
        pre: stage records to run first (e.g. WEIGHT_PACK).  want_variant: the kernel family the stage must have taken
        (s2k_program_profile_variants: 0 generic, 1 producer / consumer, 2 bf16 MFMA)."""
        from s2lc_amd import _lib

        prog = Program()
        for k, f in pre:
            prog.add(k, **f)
        prog.add(kind, **fields)
        packed = prog.pack()
        cpu = torch.zeros(self.arena.top + 256, dtype=torch.uint8)
        for name, (ref, data) in self.items.items():
            cpu[ref.off:ref.off + ref.nbytes] = data.contiguous().reshape(-1).view(torch.uint8)
        gpu = cpu.cuda()
        _lib.run(packed, _lib.Bases().set("WS", gpu), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if want_variant is not None:       # on a second copy of the inputs (a profiled run executes the stages again)
            _, var = _lib.profile_variants(packed, _lib.Bases().set("WS", cpu.cuda()), torch.cuda.current_stream().cuda_stream)
            assert int(var[-1]) == want_variant, f"{kind}: kernel family {int(var[-1])}, expected {want_variant}"
        got = gpu.cpu()
        ops_ref.run_program(packed, {WS: cpu}, D)
        for name in outputs:
            ref, _ = self.items[name]
            a = got[ref.off:ref.off + ref.nbytes].view(_DT[ref.dtype]).double()
            b = cpu[ref.off:ref.off + ref.nbytes].view(_DT[ref.dtype]).double()
            if name in sum0:
                a, b = a.view(ref.shape).sum(0), b.view(ref.shape).sum(0)
            assert torch.isfinite(b).all(), f"{kind}:{name}: oracle produced non-finite values"
            assert torch.isfinite(a).all(), f"{kind}:{name}: GPU produced non-finite values"
            denom = max(b.abs().max().item(), 1e-20)
            err = (a - b).abs().max().item() / denom
            assert err < tol, f"{kind}:{name}: rel err {err:.3e} (max |ref| {denom:.3e})"


def test_mfma_lane_maps_exact():
    """A = asymmetric small integers, B likewise: D must equal A@B exactly (guide: always check
    the C/D map with an asymmetric operand)."""
    from s2lc_amd import _lib

    a = (torch.arange(64).reshape(32, 2) % 7 - 3).float()
    b = ((torch.arange(64).reshape(2, 32) * 5) % 11 - 4).float()
    d = torch.zeros(32, 32).cuda()
    ac, bc = a.cuda(), b.cuda()
    _lib.check(_lib.lib().s2k_selftest_mfma(ac.data_ptr(), bc.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), a @ b)


# ---------------------------------------------------------------------------------------------------
# CONV (implicit GEMM)
# ---------------------------------------------------------------------------------------------------
def _conv_case(B, C1, C2, H, W, M, k, s, pt, pl, Ho, Wo, pro1, pro2, gate, bias, stats, beta=0, mode=0, flip=0,
               strides=None, seed=0, tol=1e-4, yc=None, bf16=False, res=False, scratch=False, want_variant=None, res_gelu=False):
    """bf16: the stage carries FLAG_BF16 + the bf16 weight copy, must run on the bf16 MFMA kernels (csrc/conv_bf16.hip) and is
    compared with the oracle on bf16-ROUNDED operands (f32 products and sums): what is left is f32 summation order, so the
    tolerance stays 1e-4 - rounding of the operands is arithmetic the oracle states too, not kernel error."""
    c = Case(seed)
    T = k * k
    Ct = C1 + C2
    if mode == D.MODE_GATHER2X2:
        x1 = c.t("x1", (B, C1 // 4, 2 * H, 2 * W))
    else:
        x1 = c.t("x1", (B, C1, H, W))
    x2 = c.t("x2", (B, C2, H, W)) if C2 else None
    bnv1 = c.bnv("bnv1", C1) if pro1 else None
    bnv2 = c.bnv("bnv2", C2) if (C2 and pro2) else None
    g1 = c.t("gate1", (B, C1), "rand") if gate else None
    if strides is None:
        wt = c.t("wt", (M, Ct, T), scale=(1.0 / (Ct * T)) ** 0.5)
        sm, sk, st = Ct * T, T, 1
    else:
        wshape, (sm, sk, st) = strides
        wt = c.t("wt", wshape, scale=(1.0 / (Ct * T)) ** 0.5)
    nb = M // 4 if mode == D.MODE_CONVT_SCATTER else M
    bs = c.t("bias", (nb,)) if bias else None
    YC = yc or (M // 4 if mode == D.MODE_CONVT_SCATTER else M)
    if mode == D.MODE_CONVT_SCATTER:
        y = c.t("y", (B, YC, 2 * Ho, 2 * Wo), "nan")
    else:
        y = c.t("y", (B, YC, Ho, Wo), "randn" if beta else "nan")
    nrep = D.stats_replicas(M)
    st_ref = c.t("stats", (nrep, 2, M), "zeros", "f64") if stats else None
    outs = ["y"] + (["stats"] if stats else [])
    extra = {}
    if res:
        extra["RES"] = c.t("res", (B, YC, Ho, Wo), scale=1.5)
    if res_gelu:      # Y = (conv + bias) * gelu'(RES) instead of + RES
        extra["_flags"] = D.FLAG_RES_GELU_GRAD
    if scratch:
        extra["SCRATCH"] = c.t("scratch", (8 * B * YC * Ho * Wo,), "nan")
    if want_variant is not None:
        extra["want_variant"] = want_variant
        if want_variant == 3:
            extra["_flags"] = extra.get("_flags", 0) | D.FLAG_DMA           # every tile of the LDS-DMA ring kernel, not only the shapes its routing rule takes
    if bf16:
        # An activated value within an ulp of a bf16 rounding boundary may round to the other neighbour on the GPU (its SiLU is
        # v_exp + v_rcp, its BatchNorm affine one fma): one operand then differs by 2^-8 relative.  Over a long reduction a few
        # such flips add up to ~1e-4 of the largest output; the bar for stages WITH a prologue is 1e-3 (the f32 summation-order
        # bar of 1e-4 holds for prologue-free operands, which round identically on both sides).
        if pro1 or pro2:
            tol = max(tol, 1e-3)
        pre, wp, MP, wp16 = c.pack(wt, M, Ct, T, sm, sk, st, flip, bf16=True)
        extra.update(WTB=wp16, _flags=D.FLAG_BF16, want_variant=2)
    elif want_variant == 4:
        # f32 quad copy of the weights (WEIGHT_PACK Q4_BASE), read by csrc/conv_q4.hip as WTB
        pre, wp, MP, wpq = c.pack(wt, M, Ct, T, sm, sk, st, flip, q4=True)
        extra.update(WTB=wpq, _flags=D.FLAG_Q4 | D.FLAG_DMA)      # (+ FLAG_DMA: every supported shape, not only the launcher's routing rule)
    else:
        pre, wp, MP = c.pack(wt, M, Ct, T, sm, sk, st, flip)
    c.run("CONV", outs, tol, sum0=("stats",), pre=[pre], NREP=nrep, X1=x1, BNV1=bnv1, GATE1=g1, X2=x2, BNV2=bnv2, WT=wp,
          BIAS=bs, Y=y, STATS=st_ref, B=B, C1=C1, C2=C2, H=H, W=W, M=M, KH=k, KW=k, STRIDE=s, PAD_T=pt, PAD_L=pl, HO=Ho,
          WO=Wo, PRO1=pro1, PRO2=pro2, MODE=mode, W_SM=1, W_SK=T * MP, W_ST=MP, FLIP=0, BETA=beta, YC=YC, **extra)


@pytest.mark.parametrize("B,C1,H,W,M,bias,beta,want", [
    (64, 768, 1, 52, 3072, False, 0, 1),       # fc2's data gradient of the MAE encoder (tokens as a 1 x N map): producer / consumer kernel
    (16, 512, 1, 200, 2048, True, 0, 1),
    (3, 512, 1, 200, 2048, True, 0, None),     # few tokens: whatever the launcher picks
    (2, 256, 7, 9, 96, False, 0, None),        # ragged pixels, HW % 4 != 0: scalar epilogues
    (2, 48, 16, 16, 72, True, 1, 0),           # short reduction: the generic kernel, accumulate on top
    (2, 1824, 8, 8, 304, False, 0, None),      # few pixels, deep reduction: whatever the launcher picks must honour the flag (split-K tail)
])
def test_conv1x1_times_gelu_grad_of_res(B, C1, H, W, M, bias, beta, want):
    """S2K_FLAG_RES_GELU_GRAD: Y = (conv + bias) * gelu'(RES) - fc2's data gradient through the GELU in one stage (plan/vit_plan.py
    block_bwd; timm Mlp via prithvi.py:162-183).  Kernels that do not implement it must decline the stage."""
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, 0, 0, False, bias=bias, stats=False, beta=beta, res=True, res_gelu=True,
               scratch=(C1 >= 1024), want_variant=want, seed=B + M)


@pytest.mark.parametrize("B,M,C,H,W", [(4, 240, 40, 16, 16), (3, 24, 48, 20, 28), (2, 40, 240, 64, 64), (5, 1056, 176, 8, 16), (2, 32, 16, 64, 64),
                                        (8, 1824, 304, 8, 8)])       # deep reduction: 128-channel chunks, ragged packed K
def test_dy_stored_as_bf16_between_bn_apply_and_its_1x1_readers(B, M, C, H, W):
    """bf16-mixed plans keep the dY of a BatchNorm whose only readers are bf16 1x1 stages in bf16 (opdefs CONV.X1_BF16): BN_BWD_APPLY
    rounds on store (OUT_BF16), the data-gradient CONV (X1_BF16) and the weight gradient's P operand (P_BF16) read the halves.
    M = channels of the BatchNorm (the conv's output), C = the conv's input channels."""
    HW = H * W
    # 1. the apply pass: same arithmetic as the f32 form, rounded to nearest even on store; an element within an ulp of a rounding
    #    boundary may land on the other neighbour (2^-8 relative), hence 8e-3 of the largest value
    c = Case(44)
    gp, y = c.t("gp", (B, M, HW), scale=0.3), c.t("y", (B, M, HW))
    bnv, gam = c.bnv("bnv", M), c.t("gamma", (M,), "pos")
    nrep = D.stats_replicas(M)
    st2 = c.t("st2", (nrep, 2, M), torch.randn(nrep, 2, M, dtype=torch.float64, generator=c.gen) * 3, "f64")
    dg, db = c.t("dgamma", (M,)), c.t("dbeta", (M,))
    dy = c.t("dy", (B, M, HW), "nan", "bf16")
    c.run("BN_BWD_APPLY", ["dy", "dgamma", "dbeta"], 8e-3, GP=gp, Y=y, BNV=bnv, COEF=None, DY=dy, STATS2=st2, GAMMA=gam, DGAMMA=dg, DBETA=db,
          COUNT=B * HW, B=B, C=M, HW=HW, NREP=nrep, OUT_BF16=1)
    # 2. the data gradient reads it: dX[c] = sum_m W[m][c] * dY[m]   (identical operand values on both sides: 1e-4)
    c2 = Case(45)
    dyb = c2.t("dy", (B, M, H, W), "randn", "bf16", scale=0.3)
    wt = c2.t("wt", (C, M, 1), scale=M ** -0.5)
    dx = c2.t("dx", (B, C, H, W), "randn")
    pre, wp, MP, wp16 = c2.pack(wt, C, M, 1, M, 1, 1, 0, bf16=True)
    c2.run("CONV", ["dx"], 1e-4, pre=[pre], want_variant=2, _flags=D.FLAG_BF16, WTB=wp16, X1=dyb, BNV1=None, GATE1=None, X2=None, BNV2=None,
           WT=wp, BIAS=None, Y=dx, STATS=None, B=B, C1=M, C2=0, H=H, W=W, M=C, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=W,
           PRO1=D.PRO_NONE, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=1, YC=C, NREP=1, X1_BF16=1)
    # 3. the weight gradient reads it as P: dW[m][c] += sum_pix dY[m] * Qpro[c]
    c3 = Case(46)
    dyb = c3.t("dy", (B, M, H, W), "randn", "bf16", scale=0.3)
    q, bq = c3.t("q", (B, C, H, W)), c3.bnv("bnvq", C)
    wgs = c3.t("wgs", (1, M, C), "randn")
    c3.run("WGRAD", ["wgs"], 1e-3, want_variant=2, _flags=D.FLAG_BF16, P=dyb, BNVP=None, GATEP=None, Q=q, BNVQ=bq, GATEQ=None, WGS=wgs, B=B, M=M,
           C=C, CTOT=C, H=H, W=W, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=W, PROP=D.PRO_NONE, PROQ=D.PRO_SILU, MODE=D.MODE_CONV, P_BF16=1)


@pytest.mark.parametrize("B,M,C,H,W", [(2, 64, 48, 32, 64), (3, 32, 32, 64, 64), (8, 128, 72, 16, 16), (4, 96, 64, 32, 32), (2, 40, 48, 20, 56)])
def test_dy_stored_as_bf16_read_by_3x3_stages(B, M, C, H, W):
    """the decoder's double convs: the 3x3 data-gradient CONV reads a halo element as one 16-bit load, the 3x3 weight gradient's P
    operand 8 pixels as one 16-byte load (M = channels of dY, C = the conv's input channels)"""
    c2 = Case(48)
    dyb = c2.t("dy", (B, M, H, W), "randn", "bf16", scale=0.3)
    wt = c2.t("wt", (C, M, 9), scale=(9 * M) ** -0.5)
    dx = c2.t("dx", (B, C, H, W), "randn")
    pre, wp, MP, wp16 = c2.pack(wt, C, M, 9, M * 9, 9, 1, 0, bf16=True)
    c2.run("CONV", ["dx"], 1e-4, pre=[pre], want_variant=2, _flags=D.FLAG_BF16, WTB=wp16, X1=dyb, BNV1=None, GATE1=None, X2=None, BNV2=None,
           WT=wp, BIAS=None, Y=dx, STATS=None, B=B, C1=M, C2=0, H=H, W=W, M=C, KH=3, KW=3, STRIDE=1, PAD_T=1, PAD_L=1, HO=H, WO=W,
           PRO1=D.PRO_NONE, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=9 * MP, W_ST=MP, FLIP=0, BETA=1, YC=C, NREP=1, X1_BF16=1)
    c3 = Case(49)
    dyb = c3.t("dy", (B, M, H, W), "randn", "bf16", scale=0.3)
    q, bq = c3.t("q", (B, C, H, W)), c3.bnv("bnvq", C)
    wgs = c3.t("wgs", (9, M, C), "randn")
    c3.run("WGRAD", ["wgs"], 1e-3, want_variant=2, _flags=D.FLAG_BF16, P=dyb, BNVP=None, GATEP=None, Q=q, BNVQ=bq, GATEQ=None, WGS=wgs, B=B, M=M,
           C=C, CTOT=C, H=H, W=W, KH=3, KW=3, STRIDE=1, PAD_T=1, PAD_L=1, HO=H, WO=W, PROP=D.PRO_NONE, PROQ=D.PRO_RELU, MODE=D.MODE_CONV, P_BF16=1)


def test_bf16_stored_operand_is_refused_by_the_f32_kernels():
    """nothing but the bf16 1x1 kernels reads a bf16 tensor: an unflagged stage must fail loudly, not read the halves as floats"""
    from s2lc_amd import _lib

    c = Case(47)
    B, M, C, H = 2, 16, 8, 8
    dyb = c.t("dy", (B, M, H, H), "randn", "bf16")
    q, wgs = c.t("q", (B, C, H, H)), c.t("wgs", (1, M, C), "zeros")
    with pytest.raises(_lib.S2kError, match="P_BF16"):
        c.run("WGRAD", ["wgs"], 1e-3, P=dyb, BNVP=None, GATEP=None, Q=q, BNVQ=None, GATEQ=None, WGS=wgs, B=B, M=M, C=C, CTOT=C, H=H, W=H, KH=1, KW=1,
              STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=H, PROP=D.PRO_NONE, PROQ=D.PRO_NONE, MODE=D.MODE_CONV, P_BF16=1)


# ---- bf16-mixed: the same stage records with FLAG_BF16 (csrc/conv_bf16.hip) -----------------------------------------------------------
@pytest.mark.parametrize("B,C1,H,W,M,pro,gate,bias,stats,beta", [
    (3, 24, 16, 16, 144, 0, False, False, True, 0),     # short K (24 of a 64-channel chunk), 128-row tiles
    (2, 144, 16, 16, 40, 2, True, False, True, 0),      # project conv: BatchNorm + SiLU + SE gate prologue, 64-row tiles
    (2, 40, 12, 20, 240, 0, False, False, True, 0),
    (2, 1824, 8, 8, 304, 2, True, False, True, 0),      # deep project conv: 29 chunks, few pixels
    (2, 32, 24, 24, 24, 3, False, True, False, 0),      # thin (M <= 32): 1 x 4 waves over 256 pixels, ReLU prologue, bias
    (1, 13, 8, 8, 48, 0, False, False, True, 0),        # K tail (13 channels)
    (4, 264, 64, 64, 200, 3, False, True, True, 0),     # K tail (264 = 4 x 64 + 8), ReLU prologue, bias
    (6, 288, 1, 200, 320, 0, False, True, False, 1),    # a Linear over feature-major tokens (H = 1), accumulate into Y
    (3, 256, 100, 100, 176, 1, False, False, True, 0),  # AFFINE prologue; tiles straddle images, ragged last tile
    (2, 96, 20, 20, 96, 2, False, False, True, 0),      # SiLU prologue without a gate (MBConv expand output read by ... a 1x1)
    (2, 576, 16, 16, 200, 0, False, True, True, 0),     # deep reduction, no prologue: 128-channel chunks, packed K = 9 x 64 (the last chunk's upper half lies past it)
    (3, 512, 8, 8, 2048, 0, False, False, True, 0),     # the encoder's head conv: 128-channel chunks, whole chunks only
    (2, 1056, 16, 16, 176, 2, True, False, False, 1),   # deep project conv, 128-channel chunks with a ragged K (1056 = 16.5 x 64), accumulate
])
def test_conv1x1_bf16(B, C1, H, W, M, pro, gate, bias, stats, beta):
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, pro, 0, gate, bias=bias, stats=stats, beta=beta, bf16=True)


@pytest.mark.parametrize("B,C1,C2,H,W,M,pro,beta", [
    (4, 32, 24, 64, 128, 64, 0, 0),      # (R, XW) = (2, 64): decoder concat conv (C1 a multiple of 16), two x tiles per row
    (8, 64, 0, 32, 32, 128, 3, 0),       # (4, 32): BatchNorm + ReLU prologue, zero padding after the activation
    (20, 72, 0, 16, 16, 192, 3, 1),      # (8, 16): K tail (72 = 4 x 16 + 8), accumulate
    (4, 24, 0, 30, 56, 72, 0, 0),        # (2, 56) (224-pixel inputs), odd row count, K tail
    (12, 32, 0, 28, 28, 64, 3, 0),       # (4, 28)
    (40, 64, 0, 14, 14, 128, 0, 0),      # (8, 14)
    (8, 32, 13, 64, 64, 32, 0, 0),       # thin (M <= 32): 4 x 64 pixel tiles; concat with the raw 13-band input (K tail in source 2)
    (8, 32, 0, 64, 64, 32, 3, 0),        # thin, BatchNorm + ReLU prologue
    (6, 64, 0, 62, 128, 24, 0, 1),       # thin, M = 24, ragged last tile row, accumulate
])
def test_conv3x3_bf16(B, C1, C2, H, W, M, pro, beta):
    _conv_case(B, C1, C2, H, W, M, 3, 1, 1, 1, H, W, pro, pro if C2 else 0, False, bias=True, stats=(beta == 0), beta=beta, bf16=True)


@pytest.mark.parametrize("B,C1,H,W,M,pro,gate", [
    (3, 24, 16, 16, 144, 0, False),      # expand 1x1, BM=32 config
    (2, 144, 16, 16, 40, 2, True),       # project 1x1: BN+SiLU+SE gate prologue, BM=64
    (2, 40, 12, 20, 240, 0, False),      # BM=128
    (3, 304, 7, 7, 1824, 0, False),      # 7x7 maps: ragged pixel tail, HW=49
    (2, 1824, 8, 8, 304, 2, True),       # deep project: small-problem 64x64 tiles, long K
    (2, 32, 24, 24, 4, 3, False),        # out_conv1x1: M=4
    (1, 13, 8, 8, 48, 0, False),         # K tail (13 channels)
])
def test_conv1x1(B, C1, H, W, M, pro, gate):
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, pro, 0, gate, bias=(M == 4), stats=(M != 4))


@pytest.mark.parametrize("B,C1,C2,H,W,M,pro1,pro2", [
    (2, 40, 24, 20, 20, 64, 0, 0),      # decoder concat (up-conv output + skip), BM=64
    (2, 32, 13, 32, 32, 32, 0, 0),      # input_double_conv.0: concat with the raw 13-band input
    (2, 64, 0, 14, 14, 64, 3, 0),       # second conv of a double conv: BN+ReLU prologue; 14x14 (224 path)
    (1, 88, 0, 28, 28, 128, 3, 0),      # BM=128
    (1, 32, 0, 8, 72, 32, 3, 0),        # wide rows: row segments
    (1, 16, 8, 40, 300, 32, 0, 0),      # width > tile: several x tiles per row
])
def test_conv3x3(B, C1, C2, H, W, M, pro1, pro2):
    _conv_case(B, C1, C2, H, W, M, 3, 1, 1, 1, H, W, pro1, pro2, False, bias=True, stats=True)


@pytest.mark.parametrize("B,C1,H,W,M,pro,bias,stats,beta", [
    (4, 256, 64, 64, 256, 0, False, True, 0),    # 128 x 128 tiles, BatchNorm statistics epilogue
    (4, 264, 64, 64, 200, 3, True, True, 0),     # 64-row tiles (M = 200 pads 128-row tiles by 28 %), ReLU prologue, bias, K tail (264 = 8 x 32 + 8)
    (6, 288, 1, 200, 320, 0, True, False, 1),    # a Linear over feature-major tokens (H = 1), accumulate into Y (beta)
    (3, 256, 100, 100, 176, 0, False, True, 0),  # pixel count not a multiple of the tile: tiles straddle images, ragged last tile
    (4, 512, 32, 32, 384, 0, True, True, 0),     # 96 tiles of 128 x 128 would leave CUs empty: narrow 64 x 64 tiles (384 of them)
    (6, 520, 26, 26, 512, 3, True, True, 0),     # narrow tiles, ReLU prologue, K tail, tiles straddle images (HW = 676)
])
def test_conv1x1_producer_consumer(B, C1, H, W, M, pro, bias, stats, beta):
    """shapes that take the producer / consumer kernels (csrc/igemm_pc.hip): >= 192 tiles of 128 pixels"""
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, pro, 0, False, bias=bias, stats=stats, beta=beta)


@pytest.mark.parametrize("B,C1,H,W,M,bias,stats,beta,res,scratch", [
    (32, 40, 16, 16, 240, False, True, 0, False, False),     # short reduction (K = 40 = two stages + a half stage), 256 x 128 tiles, statistics
    (8, 24, 32, 32, 144, False, True, 0, False, False),      # K = 24 (one stage + a half stage), M = 144 on 192-row tiles
    (32, 176, 16, 16, 1056, False, True, 0, False, True),    # the 16x16 expand conv: 192 x 64 tiles in three whole rounds, no split (K = 176 < 8 chunks x 4)
    (32, 1824, 8, 8, 304, False, False, 1, False, True),     # deep data gradient over 2,048 pixels: 320 x 64 tiles, K cut 8 ways, the accumulate is applied by the reduce tail
    (32, 1824, 8, 8, 304, False, True, 0, False, True),      # the same layer forward-shaped: 320 x 64 tiles, K cut 8 ways (balanced partition), statistics in the reduce tail
    (32, 3072, 8, 8, 512, True, False, 0, False, True),      # 256-row tiles x 2, K cut, bias applied by the reduce tail
    (5, 100, 10, 10, 72, True, True, 0, False, False),       # HW = 100: tiles straddle images, ragged last tile, K tail of 4 channels
    (3, 52, 1, 52, 96, True, False, 1, True, False),         # a Linear over 52-token rows (H = 1): bias + residual + accumulate, all prefetched
    (2, 768, 1, 200, 768, True, False, 0, True, True),       # ViT proj Linear: residual stream, few tokens -> K cut
    (6, 17, 4, 4, 48, False, True, 0, False, False),         # 4x4 maps (16 pixels per image), K = 17
    (2, 64, 64, 64, 40, False, False, 1, False, False),      # M = 40 on a 64-row tile, accumulate
    (16, 768, 16, 16, 128, False, False, 1, False, False),   # 128 x 768 data gradient, accumulate prefetched during the last stage (32-channel stages)
    (8, 256, 64, 64, 256, False, True, 0, False, False),     # a shape the producer / consumer kernel would take: many items per workgroup
])
def test_conv1x1_dma_ring(B, C1, H, W, M, bias, stats, beta, res, scratch):
    """prologue-free 1x1 contractions that the producer / consumer kernel leaves: the LDS-DMA ring kernel (csrc/conv_dma.hip)"""
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, 0, 0, False, bias=bias, stats=stats, beta=beta, res=res, scratch=scratch, want_variant=3)


@pytest.mark.parametrize("B,C1,H,W,M,bias,stats,beta,res,scratch", [
    (32, 40, 16, 16, 240, False, True, 0, False, False),     # K = 40 (two stages + a k-group of 8), 256 x 128 tile, statistics
    (8, 24, 32, 32, 144, False, True, 0, False, False),      # K = 24, M = 144
    (32, 176, 16, 16, 1056, False, True, 0, False, True),    # the 16x16 expand conv
    (32, 1824, 8, 8, 304, False, False, 1, False, True),     # deep data gradient over 2,048 pixels: K cut, accumulate in the reduce tail
    (32, 1824, 8, 8, 304, False, True, 0, False, True),      # forward-shaped, statistics in the reduce tail
    (32, 3072, 8, 8, 512, True, False, 0, False, True),      # bias applied by the reduce tail
    (5, 100, 10, 10, 72, True, True, 0, False, False),       # HW = 100: tiles straddle images, ragged last tile, K tail of 4 channels, bias (prefetched) + statistics
    (3, 52, 1, 52, 96, True, False, 1, True, False),         # a Linear over 52-token rows (H = 1): bias + residual + accumulate, all prefetched
    (2, 768, 1, 200, 768, True, False, 0, True, True),       # ViT proj Linear: residual stream
    (6, 17, 4, 4, 48, False, True, 0, False, False),         # 4x4 maps (16 pixels per image), K = 17
    (2, 64, 64, 64, 40, False, False, 1, False, False),      # M = 40, accumulate
    (16, 768, 16, 16, 128, False, False, 1, False, False),   # 128 x 768 data gradient, accumulate prefetched
    (8, 256, 64, 64, 256, False, True, 0, False, False),     # many items per workgroup
    (64, 768, 1, 52, 2304, True, False, 0, False, False),    # the MAE encoder's qkv Linear: 3,328 tokens
    (3, 33, 12, 12, 24, False, True, 0, False, False),       # M = 24 (smallest), K = 33
])
def test_conv1x1_quad(B, C1, H, W, M, bias, stats, beta, res, scratch):
    """prologue-free 1x1 contractions on the quad-layout kernel (csrc/conv_q4.hip): the weights' quad copy from WEIGHT_PACK, pixel-quad
    B operands, 16-byte stores straight from the accumulators"""
    _conv_case(B, C1, 0, H, W, M, 1, 1, 0, 0, H, W, 0, 0, False, bias=bias, stats=stats, beta=beta, res=res, scratch=scratch, want_variant=4)


@pytest.mark.parametrize("B,C1,C2,H,W,M,pro,beta", [
    (4, 40, 24, 64, 128, 64, 0, 0),      # (R, XW) = (2, 64): decoder concat conv, two x tiles per row
    (8, 64, 0, 32, 32, 128, 3, 0),       # (4, 32): BatchNorm + ReLU prologue, zero padding after the activation
    (20, 72, 0, 16, 16, 192, 3, 1),      # (8, 16): 64-row tiles x 3, accumulate (a data gradient on top of an existing one)
    (4, 16, 8, 30, 56, 72, 0, 0),        # (2, 56) (224-pixel inputs), odd row count, K tail (24 channels = 3 chunks)
    (12, 32, 0, 28, 28, 64, 3, 0),       # (4, 28)
    (40, 64, 0, 14, 14, 128, 0, 0),      # (8, 14)
    (32, 32, 13, 64, 64, 32, 0, 0),      # thin (M <= 32): 4 x 64 pixel tiles, consumers 1 x 4; concat with the raw input bands
    (32, 32, 0, 64, 64, 32, 3, 0),       # thin, BatchNorm + ReLU prologue
    (36, 64, 0, 62, 128, 24, 0, 1),      # thin, M = 24, ragged last tile row, accumulate
])
def test_conv3x3_producer_consumer(B, C1, C2, H, W, M, pro, beta):
    # (statistics + accumulate never occur together in a plan; the oracle takes the statistics before the accumulate)
    _conv_case(B, C1, C2, H, W, M, 3, 1, 1, 1, H, W, pro, pro if C2 else 0, False, bias=True, stats=(beta == 0), beta=beta)


@pytest.mark.parametrize("C,H,W", [(13, 32, 32), (6, 30, 26), (4, 64, 64)])
def test_stem_conv_tf_same_stride2(C, H, W):
    from s2lc_amd.plan.unet_plan import same_pads

    Ho, pt = same_pads(H, 3, 2)
    Wo, pl = same_pads(W, 3, 2)
    _conv_case(2, C, 0, H, W, 48, 3, 2, pt, pl, Ho, Wo, 0, 0, False, bias=False, stats=True)


@pytest.mark.parametrize("B,Cin,Cout,H,W,pro", [(2, 24, 16, 8, 8, 3), (1, 2048, 512, 2, 2, 2), (2, 64, 32, 16, 24, 3)])
def test_conv_transpose_scatter(B, Cin, Cout, H, W, pro):
    _conv_case(B, Cin, 0, H, W, 4 * Cout, 1, 1, 0, 0, H, W, pro, 0, False, bias=True, stats=False,
               mode=D.MODE_CONVT_SCATTER, strides=((Cin, Cout, 4), (1, 4 * Cout, 1)))


@pytest.mark.parametrize("B,Cin,Cout,H,W,pro", [(2, 24, 16, 8, 8, 3), (2, 2048, 512, 4, 4, 2), (2, 64, 32, 16, 24, 3)])
def test_conv_transpose_scatter_bf16(B, Cin, Cout, H, W, pro):
    _conv_case(B, Cin, 0, H, W, 4 * Cout, 1, 1, 0, 0, H, W, pro, 0, False, bias=True, stats=False,
               mode=D.MODE_CONVT_SCATTER, strides=((Cin, Cout, 4), (1, 4 * Cout, 1)), bf16=True)


@pytest.mark.parametrize("beta", [0, 1])
def test_conv_dgrad_3x3_flip(beta):
    # dX[c] = sum_{m,tap} W[m][c_off+c][flip tap] dY[m]: rows = a channel slice of a concat conv
    B, Mout, Ctot, c_off, Cs, H, W = 2, 64, 40, 16, 24, 12, 12
    c = Case(3)
    dy = c.t("x1", (B, Mout, H, W))
    wfull = c.t("wt_full", (Mout, Ctot, 9), scale=0.1)
    y = c.t("y", (B, Cs, H, W), "randn" if beta else "nan")
    pre, wp, MP = c.pack(wfull, Cs, Mout, 9, 9, Ctot * 9, 1, 1, src_elem_off=c_off * 9)
    c.run("CONV", ["y"], 1e-4, pre=[pre], X1=dy, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=y,
          STATS=None, B=B, C1=Mout, C2=0, H=H, W=W, M=Cs, KH=3, KW=3, STRIDE=1, PAD_T=1, PAD_L=1, HO=H, WO=W, PRO1=0,
          PRO2=0, MODE=0, W_SM=1, W_SK=9 * MP, W_ST=MP, FLIP=0, BETA=beta, YC=Cs, NREP=1)


def test_conv_dgrad_1x1_and_gather():
    _conv_case(2, 144, 0, 10, 10, 24, 1, 1, 0, 0, 10, 10, 0, 0, False, False, False, beta=1,
               strides=((144, 24, 1), (1, 24, 1)))  # dgrad of an expand conv: A[c][m] = W[m][c]
    # ConvTranspose dgrad: pseudo-channels (co,dy,dx) gathered from the 2x-resolution gradient
    _conv_case(2, 4 * 16, 0, 6, 10, 24, 1, 1, 0, 0, 6, 10, 0, 0, False, False, False, mode=D.MODE_GATHER2X2,
               strides=((24, 16 * 4, 1), (64, 1, 1)))


# ---------------------------------------------------------------------------------------------------
# WGRAD
# ---------------------------------------------------------------------------------------------------
def _wgrad_case(B, M, C, CT, c_off, H, W, k, s, pt, pl, Ho, Wo, prop, proq, gateq, mode=0, seed=0, tol=2e-4, bf16=False):
    """bf16: FLAG_BF16 - the stage must run on csrc/wgrad_bf16.hip and is compared with the oracle on bf16-ROUNDED operands
    (see _conv_case for the tolerance of stages with a prologue)."""
    c = Case(seed)
    extra = dict(_flags=D.FLAG_BF16, want_variant=2) if bf16 else {}
    if bf16 and (prop or proq):
        tol = max(tol, 1e-3)
    T = k * k
    P = c.t("p", (B, M, Ho, Wo))
    Q = c.t("q", (B, C, H, W))
    bp = c.bnv("bnvp", M) if prop else None
    bq = c.bnv("bnvq", C) if proq else None
    gq = c.t("gateq", (B, C), "rand") if gateq else None
    wgs = c.t("wgs", (T, M, CT), "randn")  # accumulates on top of existing content
    c.run("WGRAD", ["wgs"], tol, P=P, BNVP=bp, GATEP=None, Q=Q, BNVQ=bq, GATEQ=gq, WGS=wgs.at(c_off), B=B, M=M, C=C,
          CTOT=CT, H=H, W=W, KH=k, KW=k, STRIDE=s, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo, PROP=prop, PROQ=proq, MODE=mode, **extra)


@pytest.mark.parametrize("B,M,C,H,W,prop,proq,gate", [
    (5, 240, 250, 16, 16, 0, 0, False),      # 128 x 128 tiles, ragged on both sides
    (8, 240, 72, 8, 8, 0, 2, True),          # 128 x 64, SiLU + SE gate on Q (a project conv's weight gradient)
    (4, 40, 144, 16, 20, 0, 3, False),       # 64 x 128 (M = 40: one ragged 64-row tile), ReLU on Q
    (3, 64, 64, 20, 20, 2, 0, False),        # 64 x 64, SiLU prologue on P (ConvTranspose weight gradient)
    (2, 768, 384, 1, 520, 0, 0, False),      # a Linear over feature-major tokens: H = 1, W = tokens; tiles straddle images (520 = 8 x 65)
    (4, 130, 130, 16, 16, 3, 0, False),      # 64-row tiles x 3, ReLU on P
    (2, 24, 4, 64, 64, 0, 3, False),         # thin on both sides: 32 x 32 tile, the waves split 256-pixel tiles
    (2, 24, 144, 32, 32, 0, 2, True),        # 32 x 128... thin M: 32 x 64 tiles, SiLU + gate on Q
    (3, 144, 24, 20, 20, 0, 0, False),       # thin C: 64 x 32 tiles; 400 pixels per image (tiles straddle images)
])
def test_wgrad_1x1_bf16(B, M, C, H, W, prop, proq, gate):
    _wgrad_case(B, M, C, C, 0, H, W, 1, 1, 0, 0, H, W, prop, proq, gate, bf16=True)


@pytest.mark.parametrize("B,M,C,CT,c_off,H,W,proq", [
    (2, 64, 64, 64, 0, 6, 64, 3),        # (R, XW) = (2, 64), BatchNorm + ReLU on Q: zero padding after the activation
    (1, 128, 88, 88, 0, 10, 128, 0),     # (2, 64), two x tiles per row (the halo columns come from the neighbouring tile), ragged c tile
    (2, 72, 40, 104, 64, 12, 32, 3),     # (4, 32), channel slice of a concat conv (CTOT > C), ragged m / c tiles
    (2, 256, 64, 64, 0, 16, 16, 0),      # (8, 16): 4 m-tiles
    (3, 64, 128, 128, 0, 10, 16, 3),     # (8, 16), H not a multiple of R
    (1, 32, 13, 45, 32, 10, 128, 0),     # 13 of 64 columns used, channel slice of a concat conv
    (2, 24, 32, 32, 0, 5, 64, 3),        # 32 x 32 tile (waves split the pixels of 4 x 64 tiles), odd height
    (2, 64, 24, 24, 0, 9, 128, 0),       # 64 x 32 tiles, two x tiles per row, H not a multiple of 4
    (2, 32, 64, 64, 0, 8, 64, 3),        # 32 x 64 tiles
    (2, 96, 80, 80, 0, 9, 112, 3),       # (2, 56) tiles (224-pixel inputs), two x tiles per row, odd height, ragged m / c tiles
])
def test_wgrad_3x3_bf16(B, M, C, CT, c_off, H, W, proq):
    _wgrad_case(B, M, C, CT, c_off, H, W, 3, 1, 1, 1, H, W, 0, proq, False, bf16=True)


@pytest.mark.parametrize("B,M,C,H,W,proq,gate", [(2, 40, 144, 16, 16, 2, True), (3, 144, 24, 12, 12, 0, False),
                                                   (2, 4, 32, 16, 16, 3, False), (3, 200, 130, 7, 7, 0, False)])
def test_wgrad_1x1(B, M, C, H, W, proq, gate):
    _wgrad_case(B, M, C, C, 0, H, W, 1, 1, 0, 0, H, W, 0, proq, gate)


@pytest.mark.parametrize("B,M,C,CT,c_off,H,W,proq", [(2, 64, 24, 64, 40, 20, 20, 0), (2, 32, 32, 32, 0, 24, 40, 3),
                                                       (1, 32, 13, 45, 32, 32, 32, 0), (2, 128, 88, 88, 0, 14, 14, 3),
                                                       (1, 48, 40, 40, 0, 9, 130, 0)])
def test_wgrad_3x3(B, M, C, CT, c_off, H, W, proq):
    _wgrad_case(B, M, C, CT, c_off, H, W, 3, 1, 1, 1, H, W, 0, proq, False)


@pytest.mark.parametrize("B,M,C,CT,c_off,H,W,proq", [
    (2, 64, 64, 64, 0, 6, 64, 3),        # (R, XWE) = (1, 64): one row per tile
    (1, 128, 88, 88, 0, 10, 112, 0),     # (1, 64) with a ragged second x tile (48 valid columns), C not a tile multiple
    (2, 72, 40, 104, 64, 12, 32, 3),     # (2, 32), channel slice of a concat conv (CTOT > C), M / C with ragged tiles
    (2, 256, 64, 64, 0, 16, 16, 0),      # (4, 16): 4 m-tiles
    (3, 64, 128, 128, 0, 10, 15, 3),     # (4, 16) with 15 valid columns and H not a multiple of R
    (4, 64, 64, 64, 0, 8, 8, 3),         # (8, 8)
    (1, 64, 96, 96, 0, 9, 56, 0),        # (1, 56)   (224-pixel inputs: 56 / 28 / 14 maps)
    (2, 96, 64, 64, 0, 28, 28, 3),       # (2, 28)
    (3, 512, 64, 64, 0, 14, 14, 3),      # (4, 14): pixel splits over 8 m-tiles
    (2, 32, 32, 32, 0, 6, 64, 3),        # thin (a 32-channel side): 128-pixel tiles, the four consumer waves split the pixel pairs
    (1, 32, 13, 45, 32, 10, 130, 0),     # thin, 13 of 32 columns used, channel slice of a concat conv, ragged x tiles (130 = 2 x 64 + 2)
    (2, 64, 24, 24, 0, 5, 128, 0),       # thin with two m-tiles (waves 2 x 1 x 2), odd height (last tile has one row)
    (1, 24, 32, 32, 0, 4, 64, 3),        # thin, M below one tile
])
def test_wgrad_3x3_producer_consumer_tiles(B, M, C, CT, c_off, H, W, proq):
    """the shapes that take the producer / consumer kernels (csrc/wgrad_pc.hip), one case per compiled tile geometry"""
    _wgrad_case(B, M, C, CT, c_off, H, W, 3, 1, 1, 1, H, W, 0, proq, False)


@pytest.mark.parametrize("B,M,C,H,W,prop,proq,gate", [
    (5, 240, 250, 16, 16, 0, 0, False),      # 128 x 128 tiles, ragged on both sides
    (22, 240, 72, 7, 7, 0, 2, True),         # 128 x 64, SiLU + SE gate on Q, pixel count not a multiple of 64 (images straddle tiles)
    (4, 40, 144, 16, 20, 0, 3, False),       # 64 x 128... (M = 40: one ragged 64-row tile), ReLU on Q
    (3, 64, 64, 20, 20, 2, 0, False),        # 64 x 64, SiLU prologue on P (ConvTranspose weight gradient)
    (2, 768, 384, 1, 520, 0, 0, False),      # a Linear over feature-major tokens: H = 1, W = tokens
    (4, 130, 130, 16, 16, 3, 0, False),      # 64-row tiles x 3, ReLU on P
])
def test_wgrad_1x1_producer_consumer_tiles(B, M, C, H, W, prop, proq, gate):
    _wgrad_case(B, M, C, C, 0, H, W, 1, 1, 0, 0, H, W, prop, proq, gate)


def test_wgrad_stem_stride2():
    from s2lc_amd.plan.unet_plan import same_pads

    H, W = 32, 40
    Ho, pt = same_pads(H, 3, 2)
    Wo, pl = same_pads(W, 3, 2)
    _wgrad_case(2, 48, 13, 13, 0, H, W, 3, 2, pt, pl, Ho, Wo, 0, 0, False)


@pytest.mark.parametrize("B,Cin,Cout,H,W,prop", [(2, 64, 32, 8, 8, 3), (1, 200, 72, 4, 6, 2)])
def test_wgrad_conv_transpose(B, Cin, Cout, H, W, prop):
    _wgrad_case(B, Cin, Cout, Cout, 0, 2 * H, 2 * W, 2, 2, 0, 0, H, W, prop, 0, False, mode=D.MODE_GATHER2X2)


def test_wgrad_finalize():
    c = Case(5)
    entries = [(0, 8, 5, 9), (512, 12, 7, 1), (1024, 6, 4, 4)]
    table, start = [], 0
    for off, M, C, T in entries:
        table.append([off, M, C, T, start])
        start += M * C * T
    tab = c.t("table", (len(table), 5), torch.tensor(table), "i32")
    wgs = c.t("wgs", (2048,))
    grads = c.t("grads", (2048,))
    c.run("WGRAD_FINALIZE", ["grads"], 1e-6, TABLE=tab, WGS=wgs, GRADS=grads, TOTAL=start, N_ENTRIES=len(table))


# ---------------------------------------------------------------------------------------------------
# depthwise
# ---------------------------------------------------------------------------------------------------
DW_GEOS = [(2, 24, 16, 16, 3, 1), (2, 40, 16, 16, 5, 2), (3, 16, 7, 7, 5, 1), (2, 8, 40, 40, 3, 2), (1, 6, 15, 13, 5, 2),
           (1, 4, 64, 64, 5, 1), (2, 100, 8, 8, 3, 1), (1, 3, 130, 70, 3, 1),
           # small square planes at stride 1: one wave per channel walks a chunk of the batch (ragged chunks, 4 planes per pass at 8 x 8)
           (33, 12, 16, 16, 5, 1), (37, 10, 8, 8, 5, 1), (9, 6, 8, 8, 3, 1), (5, 70, 16, 16, 3, 1), (5, 10, 32, 32, 5, 1), (3, 6, 32, 32, 3, 1),
           (3, 6, 64, 64, 3, 1), (5, 9, 64, 64, 5, 1), (2, 5, 128, 128, 3, 1),      # row bands of 16 / 8 rows with halo rows from the neighbouring bands
           (3, 6, 128, 128, 3, 2), (5, 7, 64, 64, 5, 2), (9, 5, 32, 32, 3, 2), (4, 6, 32, 32, 5, 2), (6, 5, 16, 16, 3, 2)]   # stride 2 on even planes


def _dw_geo(B, C, H, W, K, S):
    from s2lc_amd.plan.unet_plan import same_pads

    Ho, pt = same_pads(H, K, S)
    Wo, pl = same_pads(W, K, S)
    return dict(B=B, C=C, H=H, W=W, K=K, STRIDE=S, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo), Ho, Wo


@pytest.mark.parametrize("geo", DW_GEOS)
@pytest.mark.parametrize("pro", [0, 2])
def test_dwconv_fwd(geo, pro):
    B, C, H, W, K, S = geo
    g, Ho, Wo = _dw_geo(*geo)
    c = Case(1)
    x = c.t("x", (B, C, H, W))
    bnv = c.bnv("bnv", C) if pro else None
    w = c.t("w", (C, K, K), scale=0.3)
    y = c.t("y", (B, C, Ho, Wo), "nan")
    nrep = D.stats_replicas(C)
    st = c.t("stats", (nrep, 2, C), "zeros", "f64")
    c.run("DWCONV_FWD", ["y", "stats"], 1e-4, sum0=("stats",), X=x, BNV=bnv, WT=w, Y=y, STATS=st, PRO=pro, NREP=nrep, **g)


@pytest.mark.parametrize("geo", DW_GEOS)
@pytest.mark.parametrize("pro,beta", [(0, 0), (0, 1), (2, 0)])
def test_dwconv_dgrad(geo, pro, beta):
    B, C, H, W, K, S = geo
    g, Ho, Wo = _dw_geo(*geo)
    c = Case(2)
    dy = c.t("dy", (B, C, Ho, Wo))
    w = c.t("w", (C, K, K), scale=0.3)
    xr = c.t("xraw", (B, C, H, W)) if pro else None
    bnv = c.bnv("bnv", C) if pro else None
    gg = c.t("g", (B, C, H, W), "randn" if beta else "nan")
    nrep = D.stats_replicas(C)
    st = c.t("stats2", (nrep, 2, C), "zeros", "f64") if pro else None
    outs = ["g"] + (["stats2"] if pro else [])
    c.run("DWCONV_DGRAD", outs, 1e-4, sum0=("stats2",), DY=dy, WT=w, XRAW=xr, BNV=bnv, G=gg, STATS2=st, PRO=pro, BETA=beta,
          NREP=nrep, **g)


@pytest.mark.parametrize("geo", DW_GEOS)
@pytest.mark.parametrize("pro", [0, 2])
def test_dwconv_wgrad(geo, pro):
    B, C, H, W, K, S = geo
    g, Ho, Wo = _dw_geo(*geo)
    c = Case(3)
    dy = c.t("dy", (B, C, Ho, Wo))
    x = c.t("x", (B, C, H, W))
    bnv = c.bnv("bnv", C) if pro else None
    dw = c.t("dw", (C, K, K), "randn")
    c.run("DWCONV_WGRAD", ["dw"], 2e-4, DY=dy, X=x, BNV=bnv, DW=dw, PRO=pro, **g)


# ---------------------------------------------------------------------------------------------------
# BatchNorm / SE / residual / reductions
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("train", [1, 0])
def test_bn_finalize(train):
    c = Case(4)
    C, n = 70, 5000
    mean = torch.randn(C, dtype=torch.float64)
    var = torch.rand(C, dtype=torch.float64) + 0.1
    nrep = 3
    full = torch.stack([mean * n, (var + mean * mean) * n])
    parts = torch.rand(nrep, 2, C, dtype=torch.float64)
    parts = parts / parts.sum(0, keepdim=True) * full          # replicas that add up to the batch sums
    stats = c.t("stats", (nrep, 2, C), parts, "f64")
    gam, bet = c.t("gamma", (C,), "pos"), c.t("beta", (C,))
    rm, rv = c.t("rm", (C,)), c.t("rv", (C,), "pos")
    bnv = c.t("bnv", (4, C), "nan")
    c.run("BN_FINALIZE", ["bnv", "rm", "rv"], 1e-5, STATS=stats if train else None, GAMMA=gam, BETA=bet, RM=rm, RV=rv,
          BNV=bnv, COUNT=n, C=C, TRAIN=train, NREP=nrep, EPS=1e-3, MOM=0.01)


@pytest.mark.parametrize("B,C,HW", [(2, 24, 256), (3, 10, 49), (1, 3, 10000), (5, 70, 64), (33, 9, 16), (4, 300, 36)])   # (<= 64: channel-per-wave kernels)
@pytest.mark.parametrize("variant", ["relu", "silu_se", "none_dc"])
def test_bn_backward_trio(B, C, HW, variant):
    c = Case(6)
    g = c.t("g", (B, C, HW))
    y = c.t("y", (B, C, HW))
    bnv = c.bnv("bnv", C)
    gout = c.t("gout", (B, C, HW), "nan")
    nrep = D.stats_replicas(C)
    st2 = c.t("stats2", (nrep, 2, C), "zeros", "f64")
    mul = c.t("mul", (B, C), "rand") if variant == "silu_se" else None
    add = c.t("add", (B, C)) if variant == "silu_se" else None
    noise = c.t("noise", (B,), "rand") if variant == "none_dc" else None
    act = {"relu": 3, "silu_se": 2, "none_dc": 0}[variant]
    c.run("BN_BWD_REDUCE", ["gout", "stats2"], 1e-4, sum0=("stats2",), G=g, Y=y, BNV=bnv, MULBC=mul, ADDBC=add, NOISE=noise,
          GOUT=gout, STATS2=st2, B=B, C=C, HW=HW, ACT=act, NREP=nrep, KEEP=0.6, ADDSCALE=1.0 / HW)
    c2 = Case(7)
    st2 = c2.t("stats2", (2, 2, C), torch.randn(2, 2, C, dtype=torch.float64) * 10, "f64")
    gam = c2.t("gamma", (C,), "pos")
    bnv = c2.bnv("bnv", C)
    dg, db = c2.t("dgamma", (C,)), c2.t("dbeta", (C,))
    coef = c2.t("coef", (3, C), "nan")
    c2.run("BN_BWD_FINALIZE", ["dgamma", "dbeta", "coef"], 1e-5, STATS2=st2, GAMMA=gam, BNV=bnv, DGAMMA=dg, DBETA=db,
           COEF=coef, COUNT=B * HW, C=C, NREP=2)
    c3 = Case(8)
    gp, y = c3.t("gp", (B, C, HW)), c3.t("y", (B, C, HW))
    bnv, coef = c3.bnv("bnv", C), c3.t("coef", (3, C))
    dy = c3.t("dy", (B, C, HW), "nan")
    c3.run("BN_BWD_APPLY", ["dy"], 1e-5, GP=gp, Y=y, BNV=bnv, COEF=coef, DY=dy, B=B, C=C, HW=HW)
    # the form the planner emits: no COEF table, FINALIZE's arithmetic inside APPLY (replica sums, dgamma / dbeta)
    for nrep in (2, 20):
        c4 = Case(9)
        gp, y = c4.t("gp", (B, C, HW)), c4.t("y", (B, C, HW))
        bnv, gam = c4.bnv("bnv", C), c4.t("gamma", (C,), "pos")
        st2 = c4.t("stats2", (nrep, 2, C), torch.randn(nrep, 2, C, dtype=torch.float64, generator=c4.gen) * 10, "f64")
        dg, db = c4.t("dgamma", (C,)), c4.t("dbeta", (C,))
        dy = c4.t("dy", (B, C, HW), "nan")
        c4.run("BN_BWD_APPLY", ["dy", "dgamma", "dbeta"], 1e-5, GP=gp, Y=y, BNV=bnv, COEF=None, DY=dy, STATS2=st2, GAMMA=gam, DGAMMA=dg,
               DBETA=db, COUNT=B * HW, B=B, C=C, HW=HW, NREP=nrep)


@pytest.mark.parametrize("B,C,HW,ident,noise", [(2, 24, 256, True, True), (3, 10, 49, False, False), (1, 3, 9000, True, False),
                                                 (9, 70, 64, True, True), (40, 5, 100, False, True)])
def test_bn_residual(B, C, HW, ident, noise):
    c = Case(9)
    y, bnv = c.t("y", (B, C, HW)), c.bnv("bnv", C)
    idt = c.t("ident", (B, C, HW)) if ident else None
    nz = c.t("noise", (B,), "rand") if noise else None
    out = c.t("xout", (B, C, HW), "nan")
    c.run("BN_RESIDUAL", ["xout"], 1e-6, Y=y, BNV=bnv, IDENT=idt, NOISE=nz, XOUT=out, B=B, C=C, HW=HW, KEEP=0.55)


@pytest.mark.parametrize("B,C,HW", [(2, 48, 256), (3, 20, 49), (1, 5, 9000), (11, 130, 64), (3, 7, 36), (3, 5, 4100 * 4)])   # (last: workgroup per plane)
def test_se_pool_bwd_reduce_channel_sum(B, C, HW):
    c = Case(10)
    y, bnv = c.t("y", (B, C, HW)), c.bnv("bnv", C)
    pool = c.t("pool", (B, C), "nan")
    c.run("SE_POOL", ["pool"], 1e-5, Y=y, BNV=bnv, POOL=pool, B=B, C=C, HW=HW, PRO=2)
    c = Case(11)
    g, y, bnv = c.t("g", (B, C, HW)), c.t("y", (B, C, HW)), c.bnv("bnv", C)
    dg = c.t("dgate", (B, C), "nan")
    c.run("SE_BWD_REDUCE", ["dgate"], 1e-4, G=g, Y=y, BNV=bnv, DGATE=dg, B=B, C=C, HW=HW, PRO=2)
    c = Case(12)
    g = c.t("g", (B, C, HW))
    out = c.t("out", (C,), "randn")
    c.run("CHANNEL_SUM", ["out"], 1e-4, G=g, OUT=out, B=B, C=C, HW=HW)


@pytest.mark.parametrize("B,C,Q", [(2, 48, 4), (3, 144, 6), (2, 3072, 128), (1, 1056, 44), (5, 768, 32), (2, 1824, 76), (3, 1100, 64),
                                   (40, 250, 10), (33, 96, 240), (32, 3840, 160)])   # (Q <= 64: one launch; B > 32: two sample chunks of the parameter-gradient tiles; Q > 224: its generic kernel)
def test_se_fc_and_backward(B, C, Q):
    c = Case(13)
    pool = c.t("pool", (B, C), "rand")
    w1, b1 = c.t("w1", (Q, C), scale=C ** -0.5), c.t("b1", (Q,))
    w2, b2 = c.t("w2", (C, Q), scale=Q ** -0.5), c.t("b2", (C,))
    hpre, gate = c.t("hpre", (B, Q), "nan"), c.t("gate", (B, C), "nan")
    c.run("SE_FC", ["hpre", "gate"], 1e-5, POOL=pool, W1=w1, B1=b1, W2=w2, B2=b2, HPRE=hpre, GATE=gate, B=B, C=C, CSQ=Q)
    c = Case(14)
    dgate, gate = c.t("dgate", (B, C)), c.t("gate", (B, C), "rand")
    hpre, pool = c.t("hpre", (B, Q)), c.t("pool", (B, C), "rand")
    w1, w2 = c.t("w1", (Q, C), scale=C ** -0.5), c.t("w2", (C, Q), scale=Q ** -0.5)
    dw1, db1, dw2, db2 = c.t("dw1", (Q, C)), c.t("db1", (Q,)), c.t("dw2", (C, Q)), c.t("db2", (C,))
    dpool, hs = c.t("dpool", (B, C), "nan"), c.t("hs", (B, Q), "nan")
    c.run("SE_FC_BWD", ["dw1", "db1", "dw2", "db2", "dpool"], 1e-4, DGATE=dgate, GATE=gate, HPRE=hpre, POOL=pool, W1=w1,
          W2=w2, DW1=dw1, DB1=db1, DW2=dw2, DB2=db2, DPOOL=dpool, HS=hs, B=B, C=C, CSQ=Q)
    # split form (what the planner emits): data gradient first, parameter gradients as a stage of their own
    c = Case(14)
    dgate, gate = c.t("dgate", (B, C)), c.t("gate", (B, C), "rand")
    hpre, pool = c.t("hpre", (B, Q)), c.t("pool", (B, C), "rand")
    w1, w2 = c.t("w1", (Q, C), scale=C ** -0.5), c.t("w2", (C, Q), scale=Q ** -0.5)
    dw1, db1, dw2, db2 = c.t("dw1", (Q, C)), c.t("db1", (Q,)), c.t("dw2", (C, Q)), c.t("db2", (C,))
    dpool, hs = c.t("dpool", (B, C), "nan"), c.t("hs", (B, Q), "nan")
    first = ("SE_FC_BWD", dict(DGATE=dgate, GATE=gate, HPRE=hpre, POOL=pool, W1=w1, W2=w2, DW1=None, DB1=None, DW2=None, DB2=None,
                               DPOOL=dpool, HS=hs, B=B, C=C, CSQ=Q))
    c.run("SE_FC_WGRAD", ["dw1", "db1", "dw2", "db2", "dpool", "hs", "dgate", "hpre"], 1e-4, pre=[first], DGP=dgate, HS=hs, DHP=hpre,
          POOL=pool, DW1=dw1, DB1=db1, DW2=dw2, DB2=db2, B=B, C=C, CSQ=Q)


def test_axpy_memset():
    c = Case(15)
    x, y = c.t("x", (10007,)), c.t("y", (10007,))
    c.run("AXPY", ["y"], 1e-7, X=x, Y=y, COUNT=10007)
    c = Case(16)
    z = c.t("z", (1000,))
    c.run("MEMSET", ["z"], 1e-7, DST=z.at(16), BYTES=4 * 500)


# ---------------------------------------------------------------------------------------------------
# loss / argmax
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,ignore,gamma,smooth,alpha,rsum", [
    (1, 0, 2.0, 0.0, False, 0), (1, 0, 0.5, 0.1, True, 0), (1, -100, 2.0, 0.0, False, 1),
    (0, 0, 0.0, 0.0, False, 0), (0, 0, 0.0, 0.1, True, 0), (0, -100, 0.0, 0.0, False, 0)])
def test_loss_fwd_bwd(mode, ignore, gamma, smooth, alpha, rsum):
    B, C, HW = 2, 4, 33 * 17
    c = Case(17)
    lg = c.t("logits", (B, C, HW), scale=2.0)
    lab = c.t("labels", (B, HW), torch.randint(0, C, (B, HW), generator=c.gen), "i64")
    al = c.t("alpha", (C,), "pos") if alpha else None
    acc = c.t("acc", (2,), "zeros", "f64")
    loss = c.t("loss", (1,), "nan")
    common = dict(B=B, C=C, HW=HW, MODE=mode, IGNORE=ignore, REDUCE_SUM=rsum, GAMMA=gamma, SMOOTH=smooth)
    c.run("LOSS_FWD", ["loss"], 1e-5, LOGITS=lg, LABELS=lab, ALPHA=al, LOSS=loss, ACC=acc, **common)
    # backward needs ACC[1] (the CE denominator): give it the true weight sum
    c2 = Case(17)
    lg = c2.t("logits", (B, C, HW), scale=2.0)
    labels = torch.randint(0, C, (B, HW), generator=c2.gen)
    lab = c2.t("labels", (B, HW), labels, "i64")
    al = c2.t("alpha", (C,), "pos") if alpha else None
    w = c2.items["alpha"][1] if alpha else torch.ones(C)
    valid = labels != ignore
    acc = c2.t("acc", (2,), torch.tensor([0.0, float(w[labels][valid].sum())], dtype=torch.float64), "f64")
    gout = c2.t("gout", (1,), torch.tensor([1.7]))
    dl = c2.t("dlogits", (B, C, HW), "nan")
    c2.run("LOSS_BWD", ["dlogits"], 1e-4, LOGITS=lg, LABELS=lab, ALPHA=al, ACC=acc, GOUT=gout, DLOGITS=dl, **common)


def test_argmax_first_max_wins():
    c = Case(18)
    B, C, HW = 2, 5, 777
    lg = torch.randn(B, C, HW, generator=c.gen).round()  # many exact ties
    lt = c.t("logits", (B, C, HW), lg)
    mask = c.t("mask", (B, HW), torch.full((B, HW), -1), "i64")
    c.run("ARGMAX", ["mask"], 1e-12, LOGITS=lt, MASK=mask, B=B, C=C, HW=HW)


@pytest.mark.parametrize("B,C,M,H,stats,beta,bias", [(32, 1824, 304, 8, True, 0, False), (32, 3072, 512, 8, True, 0, False),
                                                       (8, 1056, 176, 16, False, 1, True), (2, 320, 70, 8, True, 0, True)])
@pytest.mark.parametrize("bf16", [False, True])
def test_conv_1x1_split_k(B, C, M, H, stats, beta, bias, bf16):
    """Deep short-N layers: with a SCRATCH region the kernel cuts K into partial tiles summed in a fixed order (f32 kernels and,
    with FLAG_BF16, csrc/conv_bf16.hip)."""
    c = Case(21)
    x = c.t("x", (B, C, H, H))
    bnv = c.bnv("bnv", C)
    w = c.t("w", (M, C), scale=C ** -0.5)
    bs = c.t("bias", (M,)) if bias else None
    y = c.t("y", (B, M, H, H), "randn" if beta else "nan")
    nrep = D.stats_replicas(M)
    st = c.t("stats", (nrep, 2, M), "zeros", "f64") if stats else None
    scratch = c.t("scratch", (8 * B * M * H * H,), "nan")
    extra = {}
    if bf16:
        pre, wp, MP, wp16 = c.pack(w, M, C, 1, C, 1, 1, 0, bf16=True)
        extra = dict(WTB=wp16, _flags=D.FLAG_BF16, want_variant=2)
    else:
        pre, wp, MP = c.pack(w, M, C, 1, C, 1, 1, 0)
    c.run("CONV", ["y"] + (["stats"] if stats else []), tol=1e-3 if bf16 else 1e-4, sum0=("stats",), pre=[pre], X1=x, BNV1=bnv, GATE1=None, X2=None,
          BNV2=None, WT=wp, BIAS=bs, Y=y, STATS=st, RES=None, SCRATCH=scratch, B=B, C1=C, C2=0, H=H, W=H, M=M, KH=1, KW=1, STRIDE=1,
          PAD_T=0, PAD_L=0, HO=H, WO=H, PRO1=D.PRO_SILU, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=beta, YC=M,
          NREP=nrep, **extra)


# ---------------------------------------------------------------------------------------------------
# tensors above 2 GiB (the Prithvi head at bs 16 reads a 16 x 768 x 224 x 224 map): buffer offsets are 32-bit, so the
# kernels base their descriptors at the image of each tile; everything past the 2 GiB mark must still be read
# ---------------------------------------------------------------------------------------------------
def _big(B=14, C=768, H=224):
    assert B * C * H * H * 4 > 2 ** 31
    return B, C, H


def test_conv3x3_reads_past_2gib():
    B, C, H = _big()
    M = 8
    c = Case(31)
    x = c.t("x", (B, C, H, H), scale=0.5)
    w = c.t("w", (M, C, 9), scale=(C * 9) ** -0.5)
    y = c.t("y", (B, M, H, H), "nan")
    pre, wp, MP = c.pack(w, M, C, 9, C * 9, 9, 1, 0)
    c.run("CONV", ["y"], tol=1e-4, pre=[pre], X1=x, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=y, STATS=None, RES=None,
          B=B, C1=C, C2=0, H=H, W=H, M=M, KH=3, KW=3, STRIDE=1, PAD_T=1, PAD_L=1, HO=H, WO=H, PRO1=0, PRO2=0, MODE=D.MODE_CONV,
          W_SM=1, W_SK=9 * MP, W_ST=MP, FLIP=0, BETA=0, YC=M, NREP=1)


def test_convt_dgrad_gather_reads_past_2gib():
    B, Cout, H2 = _big()          # G = [B, Cout, 2H, 2W] with 2H = 224
    H, Cin = H2 // 2, 8
    c = Case(32)
    g = c.t("g", (B, Cout, H2, H2), scale=0.5)
    w = c.t("w", (Cin, 4 * Cout), scale=(4 * Cout) ** -0.5)
    dx = c.t("dx", (B, Cin, H, H), "nan")
    pre, wp, MP = c.pack(w, Cin, 4 * Cout, 1, 4 * Cout, 1, 1, 0)
    c.run("CONV", ["dx"], tol=1e-4, pre=[pre], X1=g, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=dx, STATS=None, RES=None,
          B=B, C1=4 * Cout, C2=0, H=H, W=H, M=Cin, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=H, PRO1=0, PRO2=0,
          MODE=D.MODE_GATHER2X2, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=Cin, NREP=1)


def test_wgrad3x3_reads_past_2gib():
    B, C, H = _big()
    M = 8
    c = Case(33)
    dy = c.t("dy", (B, M, H, H), scale=0.1)
    x = c.t("x", (B, C, H, H), scale=0.5)
    wgs = c.t("wgs", (9, M, C), "zeros")
    c.run("WGRAD", ["wgs"], tol=2e-4, P=dy, BNVP=None, GATEP=None, Q=x, BNVQ=None, GATEQ=None, WGS=wgs, B=B, M=M, C=C, CTOT=C, H=H, W=H,
          KH=3, KW=3, STRIDE=1, PAD_T=1, PAD_L=1, HO=H, WO=H, PROP=0, PROQ=0, MODE=D.MODE_CONV)


@pytest.mark.parametrize("bf16", [False, True])
def test_conv1x1_and_wgrad1x1_straddling_tiles_past_2gib(bf16):
    """H*W = 220 * 222 is no multiple of any pixel-tile width (48840 = 8 * 6105), so tiles straddle two images; the tensor
    (15 x 768 x 220 x 222 x 4 B = 2.1 GiB) does not fit 32-bit offsets from its start: the descriptors are based at the first
    image each tile touches."""
    B, C, H, W, M = 15, 768, 220, 222, 8
    assert B * C * H * W * 4 > 2 ** 31 and (H * W) % 64 != 0 and (H * W) % 8 == 0
    _conv_case(B, C, 0, H, W, M, 1, 1, 0, 0, H, W, D.PRO_NONE, 0, False, bias=False, stats=False, seed=35, bf16=bf16)
    _wgrad_case(B, M, C, C, 0, H, W, 1, 1, 0, 0, H, W, 0, 0, False, seed=36, bf16=bf16)


def test_convt_wgrad_gather_reads_past_2gib():
    B, Cout, H2 = _big()
    H, Cin = H2 // 2, 8
    c = Case(34)
    xs = c.t("x", (B, Cin, H, H), scale=0.5)
    g = c.t("g", (B, Cout, H2, H2), scale=0.1)
    wgs = c.t("wgs", (4, Cin, Cout), "zeros")
    c.run("WGRAD", ["wgs"], tol=2e-4, P=xs, BNVP=None, GATEP=None, Q=g, BNVQ=None, GATEQ=None, WGS=wgs, B=B, M=Cin, C=Cout, CTOT=Cout,
          H=H2, W=H2, KH=2, KW=2, STRIDE=2, PAD_T=0, PAD_L=0, HO=H, WO=H, PROP=0, PROQ=0, MODE=D.MODE_GATHER2X2)


@pytest.mark.parametrize("B,C,HW", [(2, 24, 256), (3, 10, 49), (1, 5, 4100), (6, 70, 64), (35, 9, 16), (4, 300, 36), (2, 6, 128 * 129)])
def test_se_bn_two_pass_stages(B, C, HW):          # (<= 64 elements: channel-per-wave kernels; >= 4096: workgroup per plane)
    """SE_BN_SUMS (one pass: dgate + the four plane sums), SE_BN_COMBINE (per-channel BatchNorm-backward sums once the SE
    factors are known) and the recomputing form of BN_BWD_APPLY; together they equal SE_BWD_REDUCE + BN_BWD_REDUCE + APPLY."""
    c = Case(21)
    g, y = c.t("g", (B, C, HW)), c.t("y", (B, C, HW))
    bnv = c.bnv("bnv", C)
    dgate, ps = c.t("dgate", (B, C), "nan"), c.t("ps", (4, B, C), "nan")
    c.run("SE_BN_SUMS", ["dgate", "ps"], 2e-4, G=g, Y=y, BNV=bnv, DGATE=dgate, PS=ps, B=B, C=C, HW=HW, ACT=D.ACT_SILU)
    c2 = Case(22)
    ps = c2.t("ps", (4, B, C), scale=5.0)
    mul, add = c2.t("mul", (B, C), "rand"), c2.t("add", (B, C))
    st2 = c2.t("st2", (2, C), "nan", "f64")
    c2.run("SE_BN_COMBINE", ["st2"], 1e-6, PS=ps, MULBC=mul, ADDBC=add, STATS2=st2, B=B, C=C, ADDSCALE=1.0 / HW)
    c3 = Case(23)
    gp, y = c3.t("gp", (B, C, HW)), c3.t("y", (B, C, HW))
    bnv, gam = c3.bnv("bnv", C), c3.t("gamma", (C,), "pos")
    mul, add = c3.t("mul", (B, C), "rand"), c3.t("add", (B, C))
    st2 = c3.t("st2", (1, 2, C), torch.randn(1, 2, C, dtype=torch.float64, generator=c3.gen) * 10, "f64")
    dg, db = c3.t("dgamma", (C,)), c3.t("dbeta", (C,))
    c3.run("BN_BWD_APPLY", ["gp", "dgamma", "dbeta"], 2e-5, GP=gp, Y=y, BNV=bnv, COEF=None, DY=gp, STATS2=st2, GAMMA=gam, DGAMMA=dg, DBETA=db,
           MULBC=mul, ADDBC=add, COUNT=B * HW, B=B, C=C, HW=HW, NREP=1, ACT=D.ACT_SILU, ADDSCALE=1.0 / HW)
    # the same with SE_BN_COMBINE folded in: APPLY forms the sums from the plane sums itself
    c4 = Case(27)
    gp, y = c4.t("gp", (B, C, HW)), c4.t("y", (B, C, HW))
    bnv, gam = c4.bnv("bnv", C), c4.t("gamma", (C,), "pos")
    mul, add = c4.t("mul", (B, C), "rand"), c4.t("add", (B, C))
    ps = c4.t("ps", (4, B, C), scale=3.0)
    dg, db = c4.t("dgamma", (C,)), c4.t("dbeta", (C,))
    c4.run("BN_BWD_APPLY", ["gp", "dgamma", "dbeta"], 2e-5, GP=gp, Y=y, BNV=bnv, COEF=None, DY=gp, STATS2=None, GAMMA=gam, DGAMMA=dg, DBETA=db,
           MULBC=mul, ADDBC=add, PS=ps, COUNT=B * HW, B=B, C=C, HW=HW, NREP=1, ACT=D.ACT_SILU, ADDSCALE=1.0 / HW)


@pytest.mark.parametrize("B,C,H,W,k,s", [(2, 13, 32, 32, 3, 2), (1, 3, 15, 22, 3, 2), (2, 4, 17, 17, 5, 2), (1, 2, 12, 20, 3, 3)])
def test_im2col_exact(B, C, H, W, k, s):
    """patch columns of a strided conv with TF-SAME padding (odd sizes: one pad row / column more at the bottom / right)"""
    from s2lc_amd.plan.unet_plan import same_pads

    Ho, pt = same_pads(H, k, s)
    Wo, pl = same_pads(W, k, s)
    c = Case(28)
    x = c.t("x", (B, C, H, W))
    y = c.t("y", (B, C * k * k, Ho, Wo), "nan")
    c.run("IM2COL", ["y"], 1e-30, X=x, Y=y, B=B, C=C, H=H, W=W, KH=k, KW=k, STRIDE=s, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo)


@pytest.mark.parametrize("B,C,H,W", [(2, 5, 6, 8), (1, 3, 7, 5), (2, 16, 28, 28)])
def test_space_to_depth_exact(B, C, H, W):
    c = Case(24)
    x = c.t("x", (B, C, 2 * H, 2 * W))
    y = c.t("y", (B, 4 * C, H, W), "nan")
    c.run("SPACE_TO_DEPTH", ["y"], 1e-30, X=x, Y=y, B=B, C=C, H=H, W=W)


@pytest.mark.parametrize("pro", [D.PRO_NONE, D.PRO_RELU, D.PRO_SILU, D.PRO_GELU])
@pytest.mark.parametrize("B,M,C,H", [(2, 48, 160, 8), (2, 130, 64, 6), (1, 24, 24, 16)])
def test_wgrad_1x1_with_prologue_on_p(B, M, C, H, pro):
    """the ConvTranspose weight gradient as a 1x1 contraction: P = the layer input with its BatchNorm / activation prologue,
    Q = the space-to-depth output gradient"""
    c = Case(25)
    P, Q = c.t("p", (B, M, H, H)), c.t("q", (B, C, H, H))
    bnv = c.bnv("bnv", M) if pro else None
    wgs = c.t("wgs", (1, M, C), "zeros")
    c.run("WGRAD", ["wgs"], 2e-4, P=P, BNVP=bnv, GATEP=None, Q=Q, BNVQ=None, GATEQ=None, WGS=wgs, B=B, M=M, C=C, CTOT=C, H=H, W=H,
          KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=H, PROP=pro, PROQ=D.PRO_NONE, MODE=D.MODE_CONV)


@pytest.mark.parametrize("M,C,pro", [(4, 32, D.PRO_RELU), (24, 24, D.PRO_NONE), (32, 13, D.PRO_SILU)])
def test_wgrad_1x1_thin_large_map(M, C, pro):
    """>= 262144 pixels and <= 32 channels on both sides: the 256-pixel-per-stage configuration (ragged last tile: 2 x 362 x 363)"""
    c = Case(26)
    B, H, W = 2, 362, 363
    P, Q = c.t("p", (B, M, H, W)), c.t("q", (B, C, H, W))
    bnv = c.bnv("bnv", C) if pro else None
    wgs = c.t("wgs", (1, M, C), "zeros")
    c.run("WGRAD", ["wgs"], 3e-4, P=P, BNVP=None, GATEP=None, Q=Q, BNVQ=bnv, GATEQ=None, WGS=wgs, B=B, M=M, C=C, CTOT=C, H=H, W=W,
          KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=H, WO=W, PROP=D.PRO_NONE, PROQ=pro, MODE=D.MODE_CONV)


# ---------------------------------------------------------------------------------------------------
# BN_FINALIZE folded into the first consumer of its {scale, shift} (opdefs.FOLD_*)
# ---------------------------------------------------------------------------------------------------
def _fold_fields(c: "Case", C: int, n: int, nrep: int):
    """Statistics replicas of a plausible batch + affine parameters + running statistics; BNV starts as NaN (it is an output)."""
    mean = torch.randn(C, dtype=torch.float64, generator=c.gen) * 0.5
    var = torch.rand(C, dtype=torch.float64, generator=c.gen) + 0.2
    full = torch.stack([mean * n, (var + mean * mean) * n])
    parts = torch.rand(nrep, 2, C, dtype=torch.float64, generator=c.gen) + 0.5
    parts = parts / parts.sum(0, keepdim=True) * full
    f = dict(FSTATS=c.t("fstats", (nrep, 2, C), parts, "f64"), FGAMMA=c.t("fgamma", (C,), "pos"), FBETA=c.t("fbeta", (C,)),
             FRM=c.t("frm", (C,)), FRV=c.t("frv", (C,), "pos"), FCOUNT=n, FNREP=nrep, FEPS=1e-3, FMOM=0.01)
    return f, c.t("bnv", (4, C), "nan")


@pytest.mark.parametrize("geo", [(2, 48, 32, 32, 3, 1), (2, 20, 17, 17, 5, 2), (3, 600, 8, 8, 5, 1), (1, 70, 64, 64, 3, 2), (4, 7, 4, 4, 3, 1),
                                 (19, 9, 16, 16, 5, 1), (21, 5, 8, 8, 3, 1)])
def test_dwconv_fwd_with_folded_bn_finalize(geo):
    B, C, H, W, K, S = geo
    g, Ho, Wo = _dw_geo(*geo)
    c = Case(31)
    x = c.t("x", (B, C, H, W))
    w = c.t("w", (C, K, K), scale=0.3)
    y = c.t("y", (B, C, Ho, Wo), "nan")
    nrep = D.stats_replicas(C)
    st = c.t("stats", (nrep, 2, C), "zeros", "f64")
    fold, bnv = _fold_fields(c, C, B * H * W, nrep)
    c.run("DWCONV_FWD", ["y", "stats", "bnv", "frm", "frv"], 1e-4, sum0=("stats",), X=x, BNV=bnv, WT=w, Y=y, STATS=st, PRO=2, NREP=nrep, **fold, **g)


@pytest.mark.parametrize("B,C,HW", [(2, 48, 256), (3, 20, 49), (1, 5, 9000), (2, 3000, 64), (7, 100, 64), (33, 9, 16), (5, 300, 196), (2, 7, 8192)])
def test_se_pool_and_bn_residual_with_folded_bn_finalize(B, C, HW):
    c = Case(32)
    y = c.t("y", (B, C, HW))
    pool = c.t("pool", (B, C), "nan")
    fold, bnv = _fold_fields(c, C, B * HW, D.stats_replicas(C))
    c.run("SE_POOL", ["pool", "bnv", "frm", "frv"], 1e-5, Y=y, BNV=bnv, POOL=pool, B=B, C=C, HW=HW, PRO=2, **fold)
    c = Case(33)
    y = c.t("y", (B, C, HW))
    idt, nz = c.t("ident", (B, C, HW)), c.t("noise", (B,), "rand")
    out = c.t("xout", (B, C, HW), "nan")
    fold, bnv = _fold_fields(c, C, B * HW, D.stats_replicas(C))
    c.run("BN_RESIDUAL", ["xout", "bnv", "frm", "frv"], 1e-5, Y=y, BNV=bnv, IDENT=idt, NOISE=nz, XOUT=out, B=B, C=C, HW=HW, KEEP=0.55, **fold)
