"""GPU parity, stage by stage, for the Prithvi MAE-ViT stages (csrc/vit.hip, csrc/attn.hip, and the GELU / residual
additions to the conv and wgrad kernels): each is launched through the C ABI on a one-record program and compared
with oracle/ops_ref.py on identical seeded bytes.  Index outputs (ranks, gather tables, masks) must be bit-exact."""
import math

import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from s2lc_amd.plan import opdefs as D
from tests.test_ops_gpu import Case

pytestmark = pytest.mark.gpu


# forward: HW % 4 == 0, HW <= 64, C >= 64 and B < 128 run the row kernel (a sample shared by channel splits), the rest the tile kernel
LN_ROW_SHAPES = [(3, 768, 52), (2, 512, 200), (9, 130, 36), (17, 64, 32), (1, 1000, 1024), (2, 96, 196), (64, 768, 52)]


@pytest.mark.parametrize("B,C,HW,eps", [(2, 32, 50, 1e-5), (2, 768, 197, 1e-5), (1, 96, 28 * 28, 1e-6), (3, 16, 64, 1e-6), (1, 5, 3, 1e-5)]
                         + [(*s, 1e-6) for s in LN_ROW_SHAPES])
def test_chan_ln_fwd(B, C, HW, eps):
    c = Case(1)
    x = c.t("x", (B, C, HW), scale=2.0)
    g, b = c.t("gamma", (C,), "pos"), c.t("beta", (C,))
    y, mr = c.t("y", (B, C, HW), "nan"), c.t("mr", (B, HW, 2), "nan")
    c.run("CHAN_LN_FWD", ["y", "mr"], tol=2e-5, X=x, GAMMA=g, BETA=b, Y=y, MR=mr, B=B, C=C, HW=HW, EPS=eps)


@pytest.mark.parametrize("B,C,HW,accum,params,extra", [(2, 32, 50, 0, True, ""), (2, 768, 197, 1, True, ""), (1, 96, 28 * 28, 1, False, ""),
                                                       (3, 16, 64, 0, True, ""), (1, 5, 3, 0, True, ""), (5, 100, 52, 1, True, ""),
                                                       (2, 768, 197, 1, True, "dxin+dsum"), (3, 40, 50, 0, False, "dsum")]
                         + [(*s, i % 2, i % 3 != 0, ["", "dxin", "dsum", "dxin+dsum"][i % 4]) for i, s in enumerate(LN_ROW_SHAPES)])
def test_chan_ln_bwd(B, C, HW, accum, params, extra):
    c = Case(2)
    xd = torch.randn(B, C, HW, generator=c.gen) * 2
    mean = xd.mean(1)
    rstd = torch.rsqrt(xd.var(1, unbiased=False) + 1e-5)
    x = c.t("x", (B, C, HW), xd)
    mr = c.t("mr", (B, HW, 2), torch.stack([mean, rstd], -1))
    dy = c.t("dy", (B, C, HW))
    g = c.t("gamma", (C,), "pos")
    dx = c.t("dx", (B, C, HW), "randn" if accum else "nan")
    dg = c.t("dgamma", (C,), "randn") if params else None
    db = c.t("dbeta", (C,), "randn") if params else None
    dxin = c.t("dxin", (B, C, HW), "randn") if "dxin" in extra else None     # out-of-place accumulate: DX = DXIN + ...
    dsum = c.t("dsum", (C,), "randn") if "dsum" in extra else None           # DSUM[c] += sum of the new DX values
    c.run("CHAN_LN_BWD", ["dx"] + (["dgamma", "dbeta"] if params else []) + (["dsum"] if dsum is not None else []), tol=1e-4, DY=dy, X=x, MR=mr,
          GAMMA=g, DX=dx, DGAMMA=dg, DBETA=db, DXIN=dxin, DSUM=dsum, B=B, C=C, HW=HW, ACCUM=1 if dxin is not None else accum)


@pytest.mark.parametrize("act", [D.ACT_GELU, D.ACT_SILU, D.ACT_RELU])
def test_act_bwd(act):
    c = Case(3)
    n = 10007
    g, x = c.t("g", (n,)), c.t("x", (n,), scale=2.0)
    c.run("ACT_BWD", ["g"], tol=1e-5, G=g, X=x, COUNT=n, ACT=act)


@pytest.mark.parametrize("act", [D.ACT_GELU, D.ACT_SILU, D.ACT_RELU])
def test_act_fwd(act):
    c = Case(3)
    n = 10007
    x, y = c.t("x", (n,), scale=2.0), c.t("y", (n,), "nan")
    c.run("ACT_FWD", ["y"], tol=1e-5, X=x, Y=y, COUNT=n, ACT=act)


@pytest.mark.parametrize("B,C,HW", [(2, 5, 64), (3, 7, 4100), (1, 300, 16), (2, 3, 8192)])
def test_act_fwd_bn_silu_gate(B, C, HW):
    """the SE-gated MBConv activation written out once: Y = SiLU(scale[c] * X + shift[c]) * GATE[b][c]"""
    c = Case(5)
    x, y = c.t("x", (B, C, HW), scale=2.0), c.t("y", (B, C, HW), "nan")
    bnv = c.t("bnv", (4, C))
    gate = c.t("gate", (B, C), "pos")
    c.run("ACT_FWD", ["y"], tol=1e-5, X=x, Y=y, BNV=bnv, GATE=gate, COUNT=B * C * HW, ACT=D.ACT_SILU, C=C, HW=HW)


def _ident_bnv(c, name, C):
    return c.t(name, (4, C), torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]))


@pytest.mark.parametrize("B,K,M,N,gelu,res", [(2, 32, 96, 50, False, False), (2, 128, 32, 197, True, True), (3, 768, 512, 50, False, True),
                                              (1, 70, 130, 17, True, False)])
def test_linear_as_conv_with_gelu_and_residual(B, K, M, N, gelu, res):
    c = Case(4)
    x = c.t("x", (B, K, N))
    w = c.t("w", (M, K), scale=K ** -0.5)
    bias = c.t("bias", (M,))
    r = c.t("res", (B, M, N)) if res else None
    y = c.t("y", (B, M, N), "nan")
    bnv = _ident_bnv(c, "ibnv", K) if gelu else None
    pre, wp, MP = c.pack(w, M, K, 1, K, 1, 1, 0)
    c.run("CONV", ["y"], tol=1e-4, pre=[pre], X1=x, BNV1=bnv, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=bias, Y=y, STATS=None, RES=r,
          B=B, C1=K, C2=0, H=1, W=N, M=M, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=N, PRO1=D.PRO_GELU if gelu else 0, PRO2=0,
          MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=M, NREP=1)


@pytest.mark.parametrize("B,K,M,N", [(2, 128, 32, 197), (3, 40, 200, 50)])
def test_linear_wgrad_with_gelu_operand(B, K, M, N):
    c = Case(5)
    dy, x = c.t("dy", (B, M, N)), c.t("x", (B, K, N))
    bnv = _ident_bnv(c, "ibnv", K)
    wgs = c.t("wgs", (1, M, K), "zeros")
    c.run("WGRAD", ["wgs"], tol=1e-4, P=dy, BNVP=None, GATEP=None, Q=x, BNVQ=bnv, GATEQ=None, WGS=wgs, B=B, M=M, C=K, CTOT=K, H=1, W=N,
          KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=N, PROP=D.PRO_NONE, PROQ=D.PRO_GELU, MODE=D.MODE_CONV)


def test_convt_wgrad_with_gelu_operand():
    c = Case(6)
    B, Cin, Cout, H = 2, 48, 40, 7
    xs, G = c.t("x", (B, Cin, H, H)), c.t("g", (B, Cout, 2 * H, 2 * H))
    bnv = _ident_bnv(c, "ibnv", Cin)
    wgs = c.t("wgs", (4, Cin, Cout), "zeros")
    c.run("WGRAD", ["wgs"], tol=1e-4, P=xs, BNVP=bnv, GATEP=None, Q=G, BNVQ=None, GATEQ=None, WGS=wgs, B=B, M=Cin, C=Cout, CTOT=Cout,
          H=2 * H, W=2 * H, KH=2, KW=2, STRIDE=2, PAD_T=0, PAD_L=0, HO=H, WO=H, PROP=D.PRO_GELU, PROQ=D.PRO_NONE, MODE=D.MODE_GATHER2X2)


# (B, heads, head dim, tokens, row stride; 0 = tokens)
ATTN = [(2, 2, 16, 17, 0), (2, 2, 8, 50, 0), (1, 3, 64, 197, 0), (1, 2, 32, 197, 0), (2, 2, 64, 50, 0), (1, 1, 64, 224, 0), (1, 2, 32, 33, 0),
        (1, 1, 5, 3, 0), (2, 2, 32, 50, 52), (1, 2, 64, 197, 200), (1, 2, 32, 197, 200), (2, 2, 16, 17, 20), (1, 1, 32, 300, 0),
        (1, 2, 64, 260, 264), (1, 1, 7, 1, 4), (1, 2, 64, 589, 592)]   # 589 = 3 frames x 196 patches + cls


def _attn_reference(qkv, B, H, HD, L, scale):
    t = qkv[..., :L].double().reshape(B, 3, H, HD, L).permute(1, 0, 2, 4, 3)
    s = (t[0] @ t[1].transpose(-2, -1)) * scale
    return (torch.softmax(s, -1) @ t[2]).permute(0, 1, 3, 2), torch.logsumexp(s, -1)   # [B,H,HD,L], [B,H,L]


@pytest.mark.parametrize("B,H,HD,L,LS", ATTN)
def test_attn_fwd(B, H, HD, L, LS):
    c = Case(7)
    S = LS or L
    qkv = c.t("qkv", (B, 3 * H * HD, S))
    o = c.t("o", (B, H * HD, S), "nan")
    lse = c.t("lse", (B, H, S), "nan")
    c.run("ATTN_FWD", ["o", "lse"], tol=1e-4, QKV=qkv, O=o, LSE=lse, B=B, HEADS=H, HD=HD, L=L, LS=LS, SCALE=HD ** -0.5)


@pytest.mark.parametrize("B,H,HD,L,LS", ATTN)
def test_attn_bwd(B, H, HD, L, LS):
    """the backward reads the forward's O and log-sum-exp back; row padding (columns L..LS-1) comes out as zeros"""
    c = Case(8)
    S = LS or L
    qd = torch.randn(B, 3 * H * HD, S, generator=c.gen)
    od, ld = _attn_reference(qd, B, H, HD, L, HD ** -0.5)
    o_full, l_full = torch.zeros(B, H, HD, S), torch.zeros(B, H, S)
    o_full[..., :L], l_full[..., :L] = od.float(), ld.float()
    qkv = c.t("qkv", (B, 3 * H * HD, S), qd)
    o = c.t("o", (B, H * HD, S), o_full.reshape(B, H * HD, S))
    lse = c.t("lse", (B, H, S), l_full)
    do = c.t("do", (B, H * HD, S))
    dqkv = c.t("dqkv", (B, 3 * H * HD, S), "nan")
    delta = c.t("delta", (B, H, S), "nan")
    c.run("ATTN_BWD", ["dqkv"], tol=2e-4, QKV=qkv, DO=do, DQKV=dqkv, O=o, LSE=lse, DELTA=delta, B=B, HEADS=H, HD=HD, L=L, LS=LS,
          SCALE=HD ** -0.5)


@pytest.mark.parametrize("B,L,keep,ties", [(3, 196, 49, False), (2, 196, 196, False), (2, 16, 4, True), (1, 588, 147, False), (2, 7, 0, False)])
def test_mae_mask_index_exact(B, L, keep, ties):
    c = Case(9)
    nz = torch.rand(B, L, generator=c.gen)
    if ties:
        nz = (nz * 4).floor() / 4
    noise = c.t("noise", (B, L), nz)
    ids = c.t("ids", (B, L), torch.full((B, L), -7), "i64")
    mask = c.t("mask", (B, L), "nan")
    enc = c.t("enc", (B, 1 + keep), torch.full((B, 1 + keep), -9), "i32")
    dec = c.t("dec", (B, 1 + L), torch.full((B, 1 + L), -9), "i32")
    c.run("MAE_MASK_INDEX", ["ids", "mask", "enc", "dec"], tol=1e-30, NOISE=noise, IDS_RESTORE=ids, MASK=mask, ENC_IDX=enc, DEC_IDX=dec,
          B=B, L=L, KEEP=keep)


@pytest.mark.parametrize("by_src,LinS,LoutS", [(0, 0, 0), (1, 0, 0), (1, 32, 24), (0, 30, 28)])
def test_token_gather(by_src, LinS, LoutS):
    """LIN_S / LOUT_S: row strides; the padding columns of OUT come out as zeros"""
    c = Case(10)
    B, C, Lin, Lout = 3, 40, 30, 23
    src = c.t("in", (B, C, LinS or Lin))
    idxv = torch.stack([torch.randperm(Lin, generator=c.gen)[:Lout] for _ in range(B)])
    idxv[:, 0] = -1
    idxv[1, 5] = -1
    idx = c.t("idx", (B, Lout), idxv, "i32")
    fill = c.t("fill", (C,))
    pos = c.t("pos", (Lin + 1 if by_src else Lout, C))
    out = c.t("out", (B, C, LoutS or Lout), "nan")
    c.run("TOKEN_GATHER", ["out"], tol=1e-6, IN=src, IDX=idx, FILL=fill, POS=pos, OUT=out, B=B, C=C, LIN=Lin, LOUT=Lout,
          POS_BY_SRC=by_src, POS_OFF=1 if by_src else 0, LIN_S=LinS, LOUT_S=LoutS)


@pytest.mark.parametrize("with_fill,LinS,LoutS", [(True, 0, 0), (False, 0, 0), (True, 52, 32), (False, 50, 30)])
def test_token_scatter(with_fill, LinS, LoutS):
    c = Case(11)
    B, C, Lin, Lout = 3, 37, 50, 29
    dout = c.t("dout", (B, C, LoutS or Lout))
    idxv = torch.stack([torch.randperm(Lin, generator=c.gen)[:Lout] for _ in range(B)])
    idxv[:, 0] = -1
    idxv[2, 9] = -1
    idx = c.t("idx", (B, Lout), idxv, "i32")
    din = c.t("din", (B, C, LinS or Lin), "nan")
    dfill = c.t("dfill", (C,), "randn") if with_fill else None
    c.run("TOKEN_SCATTER", ["din"] + (["dfill"] if with_fill else []), tol=1e-5, DOUT=dout, IDX=idx, DIN=din, DFILL=dfill, B=B, C=C, LIN=Lin,
          LOUT=Lout, LIN_S=LinS, LOUT_S=LoutS)


@pytest.mark.parametrize("B,C,T,H,P,TUB", [(2, 3, 1, 32, 8, 1), (1, 6, 1, 224, 16, 1), (2, 3, 3, 32, 8, 1), (1, 2, 4, 16, 4, 2)])
def test_patchify(B, C, T, H, P, TUB):
    c = Case(12)
    x = c.t("x", (B, C, T, H, H))
    L = (T // TUB) * (H // P) ** 2
    out = c.t("out", (B, C * TUB * P * P, L), "nan")
    c.run("PATCHIFY", ["out"], tol=1e-30, X=x, OUT=out, B=B, C=C, T=T, H=H, W=H, P=P, TUB=TUB)


@pytest.mark.parametrize("B,C,T,H,P,TUB,norm_pix", [(2, 3, 1, 32, 8, 1, 0), (2, 3, 3, 32, 8, 1, 1), (1, 6, 1, 224, 16, 1, 0), (1, 2, 4, 16, 4, 2, 1)])
def test_mae_loss_fwd_bwd(B, C, T, H, P, TUB, norm_pix):
    c = Case(13)
    L = (T // TUB) * (H // P) ** 2
    PD = TUB * P * P * C
    LP = L + 1
    pred = c.t("pred", (B, PD, LP))
    x = c.t("imgs", (B, C, T, H, H))
    mask = c.t("mask", (B, L), (torch.rand(B, L, generator=c.gen) < 0.75).float())
    loss = c.t("loss", (1,), "nan")
    acc = c.t("acc", (2,), "zeros", "f64")
    geo = dict(B=B, C=C, T=T, H=H, W=H, P=P, TUB=TUB, LP=LP, L_OFF=1, NORM_PIX=norm_pix)
    c.run("MAE_LOSS_FWD", ["loss", "acc"], tol=2e-5, PRED=pred, IMGS=x, MASK=mask, LOSS=loss, ACC=acc, **geo)
    # backward reads ACC[1] (the mask count) as the forward left it
    c2 = Case(13)
    pred = c2.t("pred", (B, PD, LP))
    x = c2.t("imgs", (B, C, T, H, H))
    mk = (torch.rand(B, L, generator=c2.gen) < 0.75).float()
    mask = c2.t("mask", (B, L), mk)
    acc = c2.t("acc", (2,), torch.tensor([0.0, mk.sum().item()]), "f64")
    gout = c2.t("gout", (1,), torch.tensor([0.7]))
    dpred = c2.t("dpred", (B, PD, LP), "nan")
    c2.run("MAE_LOSS_BWD", ["dpred"], tol=2e-5, PRED=pred, IMGS=x, MASK=mask, ACC=acc, GOUT=gout, DPRED=dpred, **geo)


@pytest.mark.parametrize("B,C,L,off,Lout", [(2, 48, 17, 1, 16), (3, 1536, 197, 1, 196), (2, 768, 50, 0, 50), (1, 5, 3, 0, 3)])
def test_transpose_cl(B, C, L, off, Lout):
    c = Case(14)
    x = c.t("x", (B, C, L))
    y = c.t("y", (B, Lout, C), "nan")
    c.run("TRANSPOSE_CL", ["y"], tol=1e-30, X=x, Y=y, B=B, C=C, L=L, L_OFF=off, LOUT=Lout)


def test_drop_gate():
    c = Case(15)
    n = 16 * 256
    u = c.t("u", (n,), "rand")
    gate = c.t("gate", (n,), "nan")
    c.run("DROP_GATE", ["gate"], tol=1e-7, U=u, GATE=gate, COUNT=n, P=0.1)


@pytest.mark.parametrize("B,C,HW", [(64, 768, 50), (7, 2304, 197), (3, 256, 512), (5, 300, 1)])
def test_channel_sum_token_rows(B, C, HW):
    """Bias gradients of the ViT Linears: short planes take the one-wave-per-channel path (no atomics)."""
    c = Case(16)
    g = c.t("g", (B, C, HW))
    out = c.t("out", (C,), "randn")
    c.run("CHANNEL_SUM", ["out"], 1e-4, G=g, OUT=out, B=B, C=C, HW=HW)


@pytest.mark.parametrize("n,C", [(100003, 4), (64 * 224 * 224, 10), (17, 2), (5000, 64)])
def test_confusion_histogram_exact(n, C):
    c = Case(17)
    pred = c.t("pred", (n,), torch.randint(-1, C + 1, (n,), generator=c.gen), "i64")
    lab = c.t("lab", (n,), torch.randint(0, C, (n,), generator=c.gen), "i64")
    hist = c.t("hist", (C * C,), torch.randint(0, 5, (C * C,), generator=c.gen), "i64")
    c.run("CONFUSION", ["hist"], tol=1e-30, PRED=pred, LABELS=lab, HIST=hist, COUNT=n, C=C)


def test_seg_metrics_accumulate_on_gpu():
    from s2lc_amd.metrics import SegMetrics, metrics_from_hist

    g = torch.Generator().manual_seed(5)
    C = 4
    m = SegMetrics(C, ignore_index=0, device="cuda:0")
    hist = torch.zeros(C, C, dtype=torch.int64)
    for _ in range(3):
        y = torch.randint(0, C, (2, 64, 64), generator=g)
        p = torch.where(torch.rand(2, 64, 64, generator=g) < 0.7, y, torch.randint(0, C, (2, 64, 64), generator=g))
        m.update(p.cuda(), y.cuda())
        hist += torch.bincount((y * C + p).reshape(-1), minlength=C * C).view(C, C)
    assert torch.equal(m.hist.view(C, C).cpu(), hist)
    out = m.compute()
    ref = metrics_from_hist(hist, 0)
    tp = hist.diag().double()
    assert abs(out["accuracy"].item() - (tp.sum() / hist.sum()).item()) < 1e-6
    iou = (tp / (hist.sum(0) + hist.sum(1) - tp).double()).mean().item()
    assert abs(out["iou"].item() - iou) < 1e-6 and abs(ref["f1"].item() - out["f1"].item()) < 1e-7
    assert out["confusion_matrix"][0].abs().sum() == 0 and torch.allclose(out["confusion_matrix"][1:].sum(1), torch.ones(C - 1))
