"""Planner for the EfficientNet-UNet hot path: emits the forward and backward stage programs.

Mirrors the data flow of /root/reference/src/modules/efficientnet_unet.py (EfficientnetUnet.forward
:125-138, EfficientNet.encode :251-263, MBConvBlock.forward :377-387) but as a static list of
fused stages over a preplanned HBM arena:

  * every conv writes its *raw* output once; BatchNorm batch statistics are accumulated in the
    conv epilogue (f64), folded to {scale, shift, mean, invstd} by BN_FINALIZE, and the
    normalise + activation (+ SE gate) is applied in the *consumer's* load prologue, so
    activations are never materialised (except the MBConv block output, which has two consumers);
  * channel concat in the decoder is two source pointers, never a copy;
  * backward is emitted explicitly (no autograd graph): per BN a reduce / finalize / apply trio,
    per conv a wgrad + dgrad pair on MFMA.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

from . import opdefs as D
from .program import Arena, Program, TRef

CAT_SIZES = {  # efficientnet_unet.py:154-165 (size[0:4]); size[4] generalised to 32 + in_channels
    "b0": [592, 296, 152, 80], "b1": [592, 296, 152, 80], "b2": [600, 304, 152, 80],
    "b3": [608, 304, 160, 88], "b4": [624, 312, 160, 88], "b5": [640, 320, 168, 88],
    "b6": [656, 328, 168, 96], "b7": [672, 336, 176, 96],
}


@dataclass
class BlockSpec:
    kernel: int
    stride: int
    cin: int
    cout: int
    expand: int
    se: int
    residual: bool

    @property
    def cexp(self) -> int:
        return self.cin * self.expand


@dataclass
class UnetSpec:
    version: str
    in_channels: int
    num_classes: int
    stem_out: int
    head_out: int
    blocks: list[BlockSpec]
    bn_momentum: float          # already flipped (1 - config value), efficientnet_unet.py:53
    bn_eps: float
    drop_connect_rate: float | None


_TUNE_WARNED = False


def tune(name: str, default: str) -> str:
    """Planner A-B switches (tools/ experiments).  Honoured only when S2K_TUNING=1 is set as well, and announced loudly:
    a leaked variable can never silently change the product's plans."""
    import os
    import sys

    global _TUNE_WARNED
    if os.environ.get("S2K_TUNING") != "1" or name not in os.environ:
        return default
    if not _TUNE_WARNED:
        print("s2k: S2K_TUNING=1 — planner tuning switches from the environment are ACTIVE (experiments only)", file=sys.stderr)
        _TUNE_WARNED = True
    return os.environ[name]


def same_pads(size: int, k: int, s: int) -> tuple[int, int]:
    """TF 'SAME' (efficientnet_unet.py:288-297): returns (out, pad_before)."""
    out = math.ceil(size / s)
    pad = max((out - 1) * s + (k - 1) + 1 - size, 0)
    return out, pad // 2


@dataclass
class Act:
    """A tensor as its consumers see it: raw NCHW data + the prologue that turns it into values."""
    raw: TRef
    C: int
    H: int
    W: int
    bnv: TRef | None = None
    pro: int = D.PRO_NONE
    gate: TRef | None = None
    needs_grad: bool = True
    grad: TRef | None = None      # grad w.r.t. the prologue-applied value, [B,C,H,W]
    grad_init: bool = False
    # SE coupling (set by the SE record's backward, consumed by the producer's BN backward)
    mulbc: TRef | None = None
    addbc: TRef | None = None
    addscale: float = 0.0
    # set when a consumer's dgrad already applied act' and accumulated the BN-backward sums
    fused_stats2: TRef | None = None
    # SiLU(BN(y)) produced by dwconv_bn: the SE backward may collect the BatchNorm-backward plane sums in its own pass
    # (SE_BN_SUMS); `se_sums` then tells the producer's backward to combine them instead of running BN_BWD_REDUCE
    bn_silu_producer: bool = False
    se_sums: TRef | None = None


@dataclass
class ParamLayout:
    params: dict = field(default_factory=dict)   # name -> (float offset, shape)
    bufs: dict = field(default_factory=dict)     # running_mean / running_var -> (float offset, shape)
    nbt: list = field(default_factory=list)      # names of num_batches_tracked, in order
    n_params: int = 0
    n_bufs: int = 0

    def add_param(self, name, shape):
        """Every parameter starts on a 256-byte boundary of the flat buffer (vector loads of weight
        rows; the padding floats are zero and stay zero under Adam)."""
        n = 1
        for s in shape:
            n *= s
        self.params[name] = (self.n_params, tuple(shape))
        self.n_params += (n + 63) // 64 * 64

    def add_buf(self, name, shape):
        n = 1
        for s in shape:
            n *= s
        self.bufs[name] = (self.n_bufs, tuple(shape))
        self.n_bufs += n


@dataclass
class UnetPlan:
    spec: UnetSpec
    B: int
    H: int
    W: int
    training: bool
    fwd: Program
    bwd: Program | None
    ws_bytes: int
    aux_bytes: int
    const_table: list           # int32 blob of the CONST base (WEIGHT_PACK / WGRAD_FINALIZE tables)
    layout: ParamLayout
    n_noise_rows: int
    bwd_param_marks: list       # backward segments (op_begin, op_end, lo, hi): after ops [begin, end) ran, the
                                # gradients of flat floats [lo, hi) are final (bucketed all-reduce can start)
    logits_shape: tuple
    tensors: dict               # debug: name -> TRef
    wpack_bytes: int = 0        # packed-weight scratch (WPACK base)


class _P:
    """Planner state."""

    def __init__(self, spec: UnetSpec, layout: ParamLayout, B: int, H: int, W: int, training: bool, want_bwd: bool | None = None):
        """training: BatchNorm on batch statistics (+ running-stat update), drop-connect / dropout active (module.train());
        want_bwd: emit the backward program (default: iff training; eval-mode plans may carry one too: torch autograd
        differentiates an eval()-mode module just the same)."""
        self.spec, self.layout, self.B, self.H, self.W, self.training = spec, layout, B, H, W, training
        self.want_bwd = training if want_bwd is None else want_bwd
        self.ws = Arena(D.BASE["WS"])
        self.aux = Arena(D.BASE["AUX"])
        self.fwd = Program("unet_fwd")
        self.bwd = Program("unet_bwd")
        self.tape: list = []
        self.table: list = []
        self.table_total = 0
        self.tensors: dict = {}
        self.marks: list = []
        self.wpack = Arena(D.BASE["WPACK"])
        self.pack_rows = {"fwd": [], "bwd": []}
        self.blob: list[int] = []      # int32 words of the CONST base (tables; float constants as their bit patterns)

    def const_table(self, rows, width) -> TRef:
        off = len(self.blob) * 4
        for r in rows:
            assert len(r) == width
            self.blob.extend(int(v) for v in r)
        while len(self.blob) % 64:
            self.blob.append(0)
        return TRef(D.BASE["CONST"], off, (len(rows), width), "i32")

    def const_floats(self, values, shape) -> TRef:
        import numpy as np

        off = len(self.blob) * 4
        self.blob.extend(int(v) for v in np.asarray(values, dtype=np.float32).reshape(-1).view(np.int32))
        while len(self.blob) % 64:
            self.blob.append(0)
        return TRef(D.BASE["CONST"], off, tuple(shape), "f32")

    def pack_weight(self, which: str, wname: str, M: int, K: int, T: int, s_m: int, s_k: int, s_t: int, flip: int,
                    src_elem_off: int = 0):
        """Schedule W -> padded K-major copy for the implicit GEMM; returns (packed ref, MP)."""
        MP = (M + 127) // 128 * 128
        KP = (K + 63) // 64 * 64
        off, _ = self.layout.params[wname]
        dst = self.wpack.alloc(f"pack:{which}:{wname}:{src_elem_off}", (KP * T, MP))
        rows = self.pack_rows[which]
        start = rows[-1][11] + rows[-1][10] * rows[-1][4] * rows[-1][9] if rows else 0
        rows.append([off + src_elem_off, dst.off // 4, M, K, T, s_m, s_k, s_t, flip, MP, KP, start])
        return dst, MP

    # -- references into the flat parameter / grad / buffer bases --------------------------
    def param(self, name) -> TRef:
        off, shape = self.layout.params[name]
        return TRef(D.BASE["PARAMS"], off * 4, shape, "f32", name)

    def pgrad(self, name) -> TRef:
        off, shape = self.layout.params[name]
        return TRef(D.BASE["GRADS"], off * 4, shape, "f32", "d:" + name)

    def wgs(self, name) -> TRef:
        off, shape = self.layout.params[name]
        return TRef(D.BASE["WGS"], off * 4, shape, "f32", "wgs:" + name)

    def buf(self, name) -> TRef:
        off, shape = self.layout.bufs[name]
        return TRef(D.BASE["BUFS"], off * 4, shape, "f32", name)

    def alloc(self, name, shape, dtype="f32") -> TRef:
        t = self.ws.alloc(name, shape, dtype)
        self.tensors[name] = t
        return t

    def grad_of(self, a: Act, name: str) -> TRef:
        if a.grad is None:
            a.grad = self.alloc("g:" + name, (self.B, a.C, a.H, a.W))
        return a.grad

    def table_entry(self, wname: str, M: int, C: int, T: int):
        off, _ = self.layout.params[wname]
        self.table.append((off, M, C, T, self.table_total))
        self.table_total += M * C * T


# ------------------------------------------------------------------------------------------
# forward emission helpers (each also pushes a backward closure on the tape)
# ------------------------------------------------------------------------------------------

def _bn_forward(p: _P, prefix: str, y: TRef, C: int, count: int, stats: TRef | None, eps: float, mom: float) -> TRef:
    bnv = p.alloc("bnv:" + prefix, (4, C))
    p.fwd.add("BN_FINALIZE", STATS=stats, GAMMA=p.param(prefix + ".weight"), BETA=p.param(prefix + ".bias"),
              RM=p.buf(prefix + ".running_mean"), RV=p.buf(prefix + ".running_var"), BNV=bnv,
              COUNT=count, C=C, TRAIN=int(p.training), NREP=D.stats_replicas(C), EPS=eps, MOM=mom)
    return bnv


def _stats(p: _P, name: str, C: int) -> TRef:
    return p.aux.alloc(name, (D.stats_replicas(C), 2, C), "f64")


def _bn_backward(p: _P, prefix: str, G: TRef, y: TRef, bnv: TRef, C: int, HW: int, act: int,
                 mulbc=None, addbc=None, addscale=0.0, noise=None, keep=1.0, inplace=True,
                 pre_stats: TRef | None = None, out_bf16: bool = False) -> TRef:
    """REDUCE -> FINALIZE -> APPLY; returns dY (grad w.r.t. the raw conv output).
    `pre_stats`: G already is g' and its sums were accumulated by the producer of G (fused dgrad).
    out_bf16 (bf16-mixed plans): dY is written as bf16 into its own tensor - its only readers are bf16 MFMA stages that would round
    exactly these values themselves (opdefs CONV.X1_BF16), so the results do not change and three passes move half the bytes."""
    B = p.B
    if pre_stats is not None:
        st2, gp = pre_stats, G
    else:
        st2 = _stats(p, "stats2:" + prefix, C)
        gp = G if inplace else p.alloc("gp:" + prefix, (B, C, HW))
        p.bwd.add("BN_BWD_REDUCE", G=G, Y=y, BNV=bnv, MULBC=mulbc, ADDBC=addbc, NOISE=noise, GOUT=gp, STATS2=st2,
                  B=B, C=C, HW=HW, ACT=act, NREP=D.stats_replicas(C), KEEP=keep, ADDSCALE=addscale)
    # the FINALIZE step (replica sums -> coefficients, dgamma / dbeta) runs inside APPLY: one launch less per BatchNorm
    dy = p.alloc("dy16:" + prefix, (B, C, HW), "bf16") if out_bf16 else gp
    p.bwd.add("BN_BWD_APPLY", GP=gp, Y=y, BNV=bnv, COEF=None, DY=dy, STATS2=st2, GAMMA=p.param(prefix + ".weight"),
              DGAMMA=p.pgrad(prefix + ".weight"), DBETA=p.pgrad(prefix + ".bias"), COUNT=B * HW, B=B, C=C, HW=HW,
              NREP=D.stats_replicas(C), EVAL=int(not p.training), OUT_BF16=int(out_bf16))
    return dy


def _conv_geometry(src: Act, k: int, stride: int, same: bool):
    if same:
        Ho, pt = same_pads(src.H, k, stride)
        Wo, pl = same_pads(src.W, k, stride)
    else:  # nn.Conv2d(k, stride 1, padding k//2)
        Ho, Wo, pt, pl = src.H, src.W, k // 2, k // 2
    return Ho, Wo, pt, pl


def _dy_bf16_ok(p: _P, srcs: list[Act], M: int, k: int, stride: int, Ho: int, Wo: int, bias_grad_from: str | None) -> bool:
    """May the dY in front of this conv's backward be STORED as bf16?  Only where every reader is a bf16 stage (1x1 or 3x3 stride 1)
    without a prologue on that operand (csrc/conv_bf16.hip X16, wgrad_bf16.hip P16): it would round the same values itself."""
    from . import bf16 as B16

    if not (getattr(p, "bf16", False) and k in (1, 3) and stride == 1 and bias_grad_from is None and Wo % 8 == 0
            and tune("S2K_DY_BF16", "1") != "0"):
        return False
    geo = dict(B=p.B, H=Ho, W=Wo, HO=Ho, WO=Wo, KH=k, KW=k, STRIDE=1, PAD_T=k // 2, PAD_L=k // 2, MODE=D.MODE_CONV)
    for s_ in srcs:
        if s_.H != Ho or s_.W != Wo:
            return False
        if not B16.wgrad_ok(dict(geo, M=M, C=s_.C, PROP=D.PRO_NONE, PROQ=s_.pro, GATEP=None, GATEQ=s_.gate)):
            return False
        if s_.needs_grad and not B16.conv_ok(dict(geo, C1=M, C2=0, M=s_.C, PRO1=D.PRO_NONE, PRO2=D.PRO_NONE, GATE1=None)):
            return False
    return True


def _conv_dgrad_wgrad(p: _P, wname: str, dY: TRef, srcs: list[Act], M: int, k: int, stride: int,
                      pt: int, pl: int, Ho: int, Wo: int, bias_grad_from: str | None):
    """Backward of a dense conv given dY [B,M,Ho,Wo] (f32, or bf16 where _dy_bf16_ok said so: dY.dtype tells)."""
    B = p.B
    T = k * k
    dy16 = int(dY.dtype == "bf16")
    Ctot = sum(s.C for s in srcs)
    # 1x1: the scratch layout [tap][M][C] IS the parameter's layout [M][C][1], so the kernel accumulates straight into the
    # (zeroed) gradient buffer: no scratch memset, no WGRAD_FINALIZE traffic for these weights (most of the encoder / every Linear)
    direct = T == 1
    if not direct:
        p.table_entry(wname, M, Ctot, T)
    wtarget = p.pgrad(wname) if direct else p.wgs(wname)
    c_off = 0
    for s in srcs:
        p.bwd.add("WGRAD", P=dY, BNVP=None, GATEP=None, Q=s.raw, BNVQ=s.bnv, GATEQ=s.gate,
                  WGS=wtarget.at(c_off), B=B, M=M, C=s.C, CTOT=Ctot, H=s.H, W=s.W, KH=k, KW=k,
                  STRIDE=stride, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo, PROP=D.PRO_NONE, PROQ=s.pro,
                  MODE=D.MODE_CONV, P_BF16=dy16)
        c_off += s.C
    if bias_grad_from is not None:
        p.bwd.add("CHANNEL_SUM", G=dY, OUT=p.pgrad(bias_grad_from), B=B, C=M, HW=Ho * Wo)
    c_off = 0
    for i, s in enumerate(srcs):
        if s.needs_grad:
            g = p.grad_of(s, f"{wname}.src{i}")
            src_dy, Hd, Wd = dY, Ho, Wo
            if stride != 1:
                # strided conv (the stem; only when the gradient w.r.t. the network input is asked for): spread dY onto the
                # input grid with zeros in between, then it is the stride-1 case below
                Hd, Wd = min(stride * Ho, s.H + k), min(stride * Wo, s.W + k)
                src_dy = p.alloc(f"up:{wname}", (B, M, Hd, Wd))
                p.bwd.add("UPSAMPLE_ZERO", X=dY, Y=src_dy, B=B, C=M, H=Ho, W=Wo, S=stride, HO=Hd, WO=Wd)
            # dX[b][c][y][x] = sum_{m,tap} W[m][c_off+c][flip(tap)] * dY[b][m][y+ky-(k-1-pt)][...]
            wp, MP = p.pack_weight("bwd", wname, s.C, M, T, T, Ctot * T, 1, 1, src_elem_off=c_off * T)
            p.bwd.add("CONV", X1=src_dy, BNV1=None, GATE1=None, X2=None, BNV2=None,
                      WT=wp, BIAS=None, Y=g, STATS=None,
                      B=B, C1=M, C2=0, H=Hd, W=Wd, M=s.C, KH=k, KW=k, STRIDE=1,
                      PAD_T=k - 1 - pt, PAD_L=k - 1 - pl, HO=s.H, WO=s.W, PRO1=D.PRO_NONE, PRO2=D.PRO_NONE,
                      MODE=D.MODE_CONV, W_SM=1, W_SK=T * MP, W_ST=MP, FLIP=0, BETA=int(s.grad_init), YC=s.C, NREP=1,
                      X1_BF16=dy16 if src_dy is dY else 0)
            s.grad_init = True
        c_off += s.C


def conv_bn(p: _P, wname: str, bnprefix: str, srcs: list[Act], M: int, k: int, stride: int, same: bool,
            act: int, eps: float, mom: float, bias: str | None = None) -> Act:
    """Dense conv (+bias) -> train/eval BatchNorm -> activation, output left virtual."""
    B = p.B
    Ho, Wo, pt, pl = _conv_geometry(srcs[0], k, stride, same)
    s0 = srcs[0]
    if (stride > 1 and k > 1 and len(srcs) == 1 and s0.pro == D.PRO_NONE and s0.gate is None and not s0.needs_grad
            and tune("S2K_STEM_IM2COL", "1") != "0"):
        # a STRIDED dense conv (the stem) as patch columns + a 1x1 contraction over C*k*k pseudo-channels (opdefs.IM2COL): the
        # parameter's [M][C][k][k] layout is the 1x1 weight [M][C*k*k], its gradient accumulates straight into the gradient buffer
        # and the columns written in the forward pass serve the weight gradient again.  (With a gradient w.r.t. the source wanted,
        # the strided path below and its zero-insertion data gradient stay.)
        col = p.alloc("col:" + wname, (B, s0.C * k * k, Ho, Wo))
        p.fwd.add("IM2COL", X=s0.raw, Y=col, B=B, C=s0.C, H=s0.H, W=s0.W, KH=k, KW=k, STRIDE=stride, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo)
        return conv_bn(p, wname, bnprefix, [Act(col, s0.C * k * k, Ho, Wo, needs_grad=False)], M, 1, 1, False, act, eps, mom, bias)
    y = p.alloc("y:" + wname, (B, M, Ho, Wo))
    stats = _stats(p, "stats:" + bnprefix, M) if p.training else None
    s1 = srcs[0]
    s2 = srcs[1] if len(srcs) > 1 else None
    Ctot = sum(s.C for s in srcs)
    T = k * k
    wp, MP = p.pack_weight("fwd", wname, M, Ctot, T, Ctot * T, T, 1, 0)
    p.fwd.add("CONV", X1=s1.raw, BNV1=s1.bnv, GATE1=s1.gate, X2=s2.raw if s2 else None,
              BNV2=s2.bnv if s2 else None, WT=wp, BIAS=p.param(bias) if bias else None, Y=y,
              STATS=stats, B=B, C1=s1.C, C2=s2.C if s2 else 0, H=s1.H, W=s1.W, M=M, KH=k, KW=k, STRIDE=stride,
              PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo, PRO1=s1.pro, PRO2=s2.pro if s2 else 0, MODE=D.MODE_CONV,
              W_SM=1, W_SK=T * MP, W_ST=MP, FLIP=0, BETA=0, YC=M, NREP=D.stats_replicas(M))
    if s2 is not None:
        assert s2.gate is None
    bnv = _bn_forward(p, bnprefix, y, M, B * Ho * Wo, stats, eps, mom)
    out = Act(y, M, Ho, Wo, bnv, act)

    def backward():
        if not out.grad_init:
            raise RuntimeError(f"no gradient reached {wname}")
        bias_g = bias if (bias and not p.training) else None
        dY = _bn_backward(p, bnprefix, out.grad, y, bnv, M, Ho * Wo, act,
                          out.mulbc, out.addbc, out.addscale, pre_stats=out.fused_stats2,
                          out_bf16=_dy_bf16_ok(p, srcs, M, k, stride, Ho, Wo, bias_g))
        # a bias in front of TRAIN-mode BatchNorm has exactly zero gradient (the batch mean absorbs it: sum of dY is 0), so no
        # stage is spent on it; with eval-mode BatchNorm (constant statistics) it is sum(dY) like any other bias
        _conv_dgrad_wgrad(p, wname, dY, srcs, M, k, stride, pt, pl, Ho, Wo, bias_g)

    p.tape.append(backward)
    return out


def project_conv_bn_residual(p: _P, idx: int, wname: str, bnprefix: str, src: Act, M: int, ident: Act | None,
                             dc_rate: float | None, eps: float, mom: float, out_ref: TRef | None = None) -> Act:
    """MBConv tail: 1x1 project -> BN -> [drop-connect] + identity, materialised (it has two consumers).
    out_ref: where to materialise it (EfficientNet.encode returns some block outputs: they are written straight into OUT)."""
    B = p.B
    H, W = src.H, src.W
    y = p.alloc("y:" + wname, (B, M, H, W))
    stats = _stats(p, "stats:" + bnprefix, M) if p.training else None
    wp, MP = p.pack_weight("fwd", wname, M, src.C, 1, src.C, 1, 1, 0)
    csrc = src          # what the conv (and its weight gradient) read
    if (not getattr(p, "bf16", False) and src.gate is not None and src.pro == D.PRO_SILU and (H * W) % 4 == 0 and src.C >= 256
            and tune("S2K_SE_MATERIALIZE", "0") == "1"):
        # Experiment (off by default, S2K_TUNING=1 S2K_SE_MATERIALIZE=1): SiLU(BN(y)) * gate written out once, so that the project conv and
        # its weight gradient read a plain tensor on the producer / consumer kernels instead of re-evaluating the 8-instruction prologue
        # per output-channel tile.  Measured on b5 13x256x256 bs 32 (gpurun_out/sem_{on,off}.json): WGRAD 9.71 -> 9.25 ms, CONV 16.15 ->
        # 16.04 ms, the extra pass 0.46 ms - 980 vs 984 tiles/s, no gain (the weight gradients already hide on the side stream).
        a = p.alloc("a:" + wname, (B, src.C, H, W))
        p.fwd.add("ACT_FWD", X=src.raw, Y=a, BNV=src.bnv, GATE=src.gate, COUNT=B * src.C * H * W, ACT=D.ACT_SILU, C=src.C, HW=H * W)
        csrc = Act(a, src.C, H, W, needs_grad=src.needs_grad)
    p.fwd.add("CONV", X1=csrc.raw, BNV1=csrc.bnv, GATE1=csrc.gate, X2=None, BNV2=None, WT=wp, BIAS=None,
              Y=y, STATS=stats, B=B, C1=src.C, C2=0, H=H, W=W, M=M, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0,
              HO=H, WO=W, PRO1=csrc.pro, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=M,
              NREP=D.stats_replicas(M))
    bnv = _bn_forward(p, bnprefix, y, M, B * H * W, stats, eps, mom)
    xout = out_ref if out_ref is not None else p.alloc(f"x:block{idx}", (B, M, H, W))
    use_dc = bool(ident is not None and dc_rate and p.training)
    keep = 1.0 - dc_rate if use_dc else 1.0
    noise = TRef(D.BASE["NOISE"], idx * B * 4, (B,), "f32", f"noise{idx}") if use_dc else None
    p.fwd.add("BN_RESIDUAL", Y=y, BNV=bnv, IDENT=ident.raw if ident is not None else None, NOISE=noise,
              XOUT=xout, B=B, C=M, HW=H * W, KEEP=keep)
    if ident is not None:
        assert ident.pro == D.PRO_NONE and ident.C == M
    out = Act(xout, M, H, W)

    def backward():
        if not out.grad_init:
            raise RuntimeError(f"no gradient reached block {idx}")
        G = out.grad
        if ident is not None and ident.needs_grad:
            if not ident.grad_init:
                ident.grad, ident.grad_init = G, True      # alias: later dgrads accumulate in place
            else:
                p.bwd.add("AXPY", X=G, Y=ident.grad, COUNT=B * M * H * W)
        dY = _bn_backward(p, bnprefix, G, y, bnv, M, H * W, D.ACT_NONE, noise=noise, keep=keep, inplace=False,
                          out_bf16=_dy_bf16_ok(p, [src], M, 1, 1, H, W, None))
        _conv_dgrad_wgrad(p, wname, dY, [csrc], M, 1, 1, 0, 0, H, W, None)
        if csrc is not src:       # the gradient w.r.t. the conv's input IS the gradient w.r.t. the gated activation the SE backward expects
            assert not src.grad_init
            src.grad, src.grad_init = csrc.grad, csrc.grad_init

    p.tape.append(backward)
    return out


def dwconv_bn(p: _P, wname: str, bnprefix: str, src: Act, k: int, stride: int, eps: float, mom: float) -> Act:
    B, C = p.B, src.C
    Ho, pt = same_pads(src.H, k, stride)
    Wo, pl = same_pads(src.W, k, stride)
    y = p.alloc("y:" + wname, (B, C, Ho, Wo))
    stats = _stats(p, "stats:" + bnprefix, C) if p.training else None
    assert src.gate is None
    geo = dict(B=B, C=C, H=src.H, W=src.W, K=k, STRIDE=stride, PAD_T=pt, PAD_L=pl, HO=Ho, WO=Wo, PRO=src.pro)
    nrep = D.stats_replicas(C)
    p.fwd.add("DWCONV_FWD", X=src.raw, BNV=src.bnv, WT=p.param(wname), Y=y, STATS=stats, NREP=nrep, **geo)
    bnv = _bn_forward(p, bnprefix, y, C, B * Ho * Wo, stats, eps, mom)
    out = Act(y, C, Ho, Wo, bnv, D.PRO_SILU, bn_silu_producer=True)

    def backward():
        if out.se_sums is not None:
            # two-pass form (csrc/ew.hip, SE_BN_SUMS): the plane sums were collected by the SE backward's pass; combine them
            # with the SE result and let APPLY recompute g' = (d * gate + dpool / HW) * silu'(u) while it streams d and y
            HWo = Ho * Wo
            dY = out.grad
            fold_combine = tune("S2K_SE_COMBINE_IN_APPLY", "1") != "0"     # SE_BN_COMBINE's arithmetic inside APPLY: one launch less
            st2 = None
            if not fold_combine:
                st2 = p.aux.alloc("stats2c:" + bnprefix, (1, 2, C), "f64")
                p.bwd.add("SE_BN_COMBINE", PS=out.se_sums, MULBC=out.mulbc, ADDBC=out.addbc, STATS2=st2, B=B, C=C, ADDSCALE=out.addscale)
            p.bwd.add("BN_BWD_APPLY", GP=dY, Y=y, BNV=bnv, COEF=None, DY=dY, STATS2=st2, GAMMA=p.param(bnprefix + ".weight"),
                      DGAMMA=p.pgrad(bnprefix + ".weight"), DBETA=p.pgrad(bnprefix + ".bias"), MULBC=out.mulbc, ADDBC=out.addbc,
                      PS=out.se_sums if fold_combine else None,
                      COUNT=B * HWo, B=B, C=C, HW=HWo, NREP=1, ACT=D.ACT_SILU, ADDSCALE=out.addscale, EVAL=int(not p.training))
        else:
            dY = _bn_backward(p, bnprefix, out.grad, y, bnv, C, Ho * Wo, D.ACT_SILU, out.mulbc, out.addbc, out.addscale)
        p.bwd.add("DWCONV_WGRAD", DY=dY, X=src.raw, BNV=src.bnv, DW=p.pgrad(wname), **geo)
        if not src.needs_grad:
            return
        if src.pro == D.PRO_NONE:
            g = p.grad_of(src, wname + ".src")
            p.bwd.add("DWCONV_DGRAD", DY=dY, WT=p.param(wname), XRAW=None, BNV=None, G=g, STATS2=None,
                      BETA=int(src.grad_init), NREP=1, **geo)
            src.grad_init = True
        else:
            # fused: dgrad * act'(u) and the BN-backward statistics of the producer; the producer's
            # backward then only runs FINALIZE + APPLY on this buffer.
            g = p.grad_of(src, wname + ".src")
            assert not src.grad_init
            st2 = _stats(p, "stats2f:" + bnprefix, C)
            p.bwd.add("DWCONV_DGRAD", DY=dY, WT=p.param(wname), XRAW=src.raw, BNV=src.bnv, G=g, STATS2=st2,
                      BETA=0, NREP=nrep, **geo)
            src.grad_init = True
            src.fused_stats2 = st2

    p.tape.append(backward)
    return out


def squeeze_excite(p: _P, prefix: str, a: Act, se: int) -> Act:
    """x * sigmoid(SE(x)) (efficientnet_unet.py:381): attaches the gate to the virtual activation."""
    B, C, HW = p.B, a.C, a.H * a.W
    pool = p.alloc("pool:" + prefix, (B, C))
    hpre = p.alloc("hpre:" + prefix, (B, se))
    gate = p.alloc("gate:" + prefix, (B, C))
    p.fwd.add("SE_POOL", Y=a.raw, BNV=a.bnv, POOL=pool, B=B, C=C, HW=HW, PRO=a.pro)
    w1, b1 = prefix + ".1.weight", prefix + ".1.bias"
    w2, b2 = prefix + ".3.weight", prefix + ".3.bias"
    p.fwd.add("SE_FC", POOL=pool, W1=p.param(w1), B1=p.param(b1), W2=p.param(w2), B2=p.param(b2),
              HPRE=hpre, GATE=gate, B=B, C=C, CSQ=se)
    a.gate = gate

    def backward():
        dgate = p.alloc("dgate:" + prefix, (B, C))
        dpool = p.alloc("dpool:" + prefix, (B, C))
        hs = p.alloc("hs:" + prefix, (B, se))
        if a.bn_silu_producer and a.pro == D.PRO_SILU and tune("S2K_SE_BN_TWO_PASS", "1") != "0":
            a.se_sums = p.alloc("se_sums:" + prefix, (4, B, C))
            p.bwd.add("SE_BN_SUMS", G=a.grad, Y=a.raw, BNV=a.bnv, DGATE=dgate, PS=a.se_sums, B=B, C=C, HW=HW, ACT=D.ACT_SILU)
        else:
            p.bwd.add("SE_BWD_REDUCE", G=a.grad, Y=a.raw, BNV=a.bnv, DGATE=dgate, B=B, C=C, HW=HW, PRO=a.pro)
        # the data-gradient part stays on the critical chain; the parameter gradients (a third of the stage's time, nothing
        # downstream reads them before the bucket is finalised) go to the side stream as their own stage
        p.bwd.add("SE_FC_BWD", DGATE=dgate, GATE=gate, HPRE=hpre, POOL=pool, W1=p.param(w1), W2=p.param(w2),
                  DW1=None, DB1=None, DW2=None, DB2=None, DPOOL=dpool, HS=hs, B=B, C=C, CSQ=se)
        p.bwd.add("SE_FC_WGRAD", DGP=dgate, HS=hs, DHP=hpre, POOL=pool, DW1=p.pgrad(w1), DB1=p.pgrad(b1), DW2=p.pgrad(w2),
                  DB2=p.pgrad(b2), B=B, C=C, CSQ=se)
        a.mulbc, a.addbc, a.addscale = gate, dpool, 1.0 / HW

    p.tape.append(backward)
    return a


def conv_transpose(p: _P, wname: str, bname: str, src: Act, Cout: int) -> Act:
    """nn.ConvTranspose2d(k=2, s=2, bias) as one GEMM with M = (co,dy,dx) and an interleaved store."""
    B, Cin, H, W = p.B, src.C, src.H, src.W
    u = p.alloc("u:" + wname, (B, Cout, 2 * H, 2 * W))
    wp, MP = p.pack_weight("fwd", wname, 4 * Cout, Cin, 1, 1, 4 * Cout, 1, 0)
    p.fwd.add("CONV", X1=src.raw, BNV1=src.bnv, GATE1=src.gate, X2=None, BNV2=None, WT=wp,
              BIAS=p.param(bname), Y=u, STATS=None, B=B, C1=Cin, C2=0, H=H, W=W, M=4 * Cout, KH=1, KW=1, STRIDE=1,
              PAD_T=0, PAD_L=0, HO=H, WO=W, PRO1=src.pro, PRO2=0, MODE=D.MODE_CONVT_SCATTER,
              W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=Cout, NREP=1)
    out = Act(u, Cout, 2 * H, 2 * W)

    def backward():
        G = out.grad
        if tune("S2K_CONVT_S2D", "1") != "0":
            # regroup the output gradient once (space-to-depth: [B][Cout][2H][2W] -> [B][(co,dy,dx)][H][W], one HBM pass), then the
            # weight gradient W[ci][(co,dy,dx)] and the data gradient are plain 1x1 contractions over 4*Cout channels on the fast
            # pixel kernels; the 2x2-gather wgrad keeps four accumulator tiles per wave and ran at ~37 TF/s (25 ms of the 158 ms
            # Prithvi segmentation step)
            G4 = p.alloc("s2d:" + wname, (B, 4 * Cout, H, W))
            p.bwd.add("SPACE_TO_DEPTH", X=G, Y=G4, B=B, C=Cout, H=H, W=W)
            p.bwd.add("WGRAD", P=src.raw, BNVP=src.bnv, GATEP=src.gate, Q=G4, BNVQ=None, GATEQ=None, WGS=p.pgrad(wname),   # [Cin][(co,dy,dx)] = the weight's own layout
                      B=B, M=Cin, C=4 * Cout, CTOT=4 * Cout, H=H, W=W, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0,
                      HO=H, WO=W, PROP=src.pro, PROQ=D.PRO_NONE, MODE=D.MODE_CONV)
            p.bwd.add("CHANNEL_SUM", G=G, OUT=p.pgrad(bname), B=B, C=Cout, HW=4 * H * W)
            if src.needs_grad:
                g = p.grad_of(src, wname + ".src")
                wp, MP = p.pack_weight("bwd", wname, Cin, 4 * Cout, 1, 4 * Cout, 1, 1, 0)
                p.bwd.add("CONV", X1=G4, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=g,
                          STATS=None, B=B, C1=4 * Cout, C2=0, H=H, W=W, M=Cin, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0,
                          HO=H, WO=W, PRO1=D.PRO_NONE, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP,
                          FLIP=0, BETA=int(src.grad_init), YC=Cin, NREP=1)
                src.grad_init = True
            return
        p.table_entry(wname, Cin, Cout, 4)
        p.bwd.add("WGRAD", P=src.raw, BNVP=src.bnv, GATEP=src.gate, Q=G, BNVQ=None, GATEQ=None, WGS=p.wgs(wname),
                  B=B, M=Cin, C=Cout, CTOT=Cout, H=2 * H, W=2 * W, KH=2, KW=2, STRIDE=2, PAD_T=0, PAD_L=0,
                  HO=H, WO=W, PROP=src.pro, PROQ=D.PRO_NONE, MODE=D.MODE_GATHER2X2)
        p.bwd.add("CHANNEL_SUM", G=G, OUT=p.pgrad(bname), B=B, C=Cout, HW=4 * H * W)
        if src.needs_grad:
            g = p.grad_of(src, wname + ".src")
            # dX[b][ci][y][x] = sum_{k=(co,dy,dx)} W[ci][k] * G[b][co][2y+dy][2x+dx]
            wp, MP = p.pack_weight("bwd", wname, Cin, 4 * Cout, 1, 4 * Cout, 1, 1, 0)
            p.bwd.add("CONV", X1=G, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=g,
                      STATS=None, B=B, C1=4 * Cout, C2=0, H=H, W=W, M=Cin, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0,
                      HO=H, WO=W, PRO1=D.PRO_NONE, PRO2=0, MODE=D.MODE_GATHER2X2, W_SM=1, W_SK=MP, W_ST=MP,
                      FLIP=0, BETA=int(src.grad_init), YC=Cin, NREP=1)
            src.grad_init = True

    p.tape.append(backward)
    return out


def out_conv(p: _P, wname: str, bname: str, src: Act, M: int) -> TRef:
    B, H, W = p.B, src.H, src.W
    logits = TRef(D.BASE["OUT"], 0, (B, M, H, W), "f32", "logits")
    wp, MP = p.pack_weight("fwd", wname, M, src.C, 1, src.C, 1, 1, 0)
    p.fwd.add("CONV", X1=src.raw, BNV1=src.bnv, GATE1=None, X2=None, BNV2=None, WT=wp,
              BIAS=p.param(bname), Y=logits, STATS=None, B=B, C1=src.C, C2=0, H=H, W=W, M=M, KH=1, KW=1, STRIDE=1,
              PAD_T=0, PAD_L=0, HO=H, WO=W, PRO1=src.pro, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP,
              FLIP=0, BETA=0, YC=M, NREP=1)

    def backward():
        dY = TRef(D.BASE["DOUT"], 0, (B, M, H, W), "f32", "dlogits")
        _conv_dgrad_wgrad(p, wname, dY, [src], M, 1, 1, 0, 0, H, W, bname)

    p.tape.append(backward)
    return logits


# ------------------------------------------------------------------------------------------

def build_layout(spec: UnetSpec) -> ParamLayout:
    """Flat parameter / buffer layout in the reference's registration order."""
    L = ParamLayout()

    def bn(prefix, c):
        L.add_param(prefix + ".weight", (c,))
        L.add_param(prefix + ".bias", (c,))
        L.add_buf(prefix + ".running_mean", (c,))
        L.add_buf(prefix + ".running_var", (c,))
        L.nbt.append(prefix + ".num_batches_tracked")

    L.add_param("encoder.stem.0.weight", (spec.stem_out, spec.in_channels, 3, 3))
    bn("encoder.stem.1", spec.stem_out)
    for i, b in enumerate(spec.blocks):
        pre = f"encoder.blocks.{i}."
        j = 0
        if b.expand != 1:
            L.add_param(pre + "stem.0.weight", (b.cexp, b.cin, 1, 1))
            bn(pre + "stem.1", b.cexp)
            j = 3
        L.add_param(pre + f"stem.{j}.weight", (b.cexp, 1, b.kernel, b.kernel))
        bn(pre + f"stem.{j + 1}", b.cexp)
        L.add_param(pre + "squeeze_excitation.1.weight", (b.se, b.cexp, 1, 1))
        L.add_param(pre + "squeeze_excitation.1.bias", (b.se,))
        L.add_param(pre + "squeeze_excitation.3.weight", (b.cexp, b.se, 1, 1))
        L.add_param(pre + "squeeze_excitation.3.bias", (b.cexp,))
        L.add_param(pre + "final_layer.0.weight", (b.cout, b.cexp, 1, 1))
        bn(pre + "final_layer.1", b.cout)
    L.add_param("encoder.conv_head.0.weight", (spec.head_out, spec.blocks[-1].cout, 1, 1))
    bn("encoder.conv_head.1", spec.head_out)
    L.add_param("encoder.fc.3.weight", (spec.num_classes, spec.head_out))
    L.add_param("encoder.fc.3.bias", (spec.num_classes,))
    ups_in, ups_out = [spec.head_out, 512, 256, 128], [512, 256, 128, 64]
    cat = CAT_SIZES[spec.version]
    for i in range(4):
        L.add_param(f"up_convs.{i}.weight", (ups_in[i], ups_out[i], 2, 2))
        L.add_param(f"up_convs.{i}.bias", (ups_out[i],))
    for i in range(4):
        pre = f"double_convs.{i}"
        L.add_param(pre + ".0.weight", (ups_out[i], cat[i], 3, 3))
        L.add_param(pre + ".0.bias", (ups_out[i],))
        bn(pre + ".1", ups_out[i])
        L.add_param(pre + ".3.weight", (ups_out[i], ups_out[i], 3, 3))
        L.add_param(pre + ".3.bias", (ups_out[i],))
        bn(pre + ".4", ups_out[i])
    L.add_param("input_up_conv.weight", (64, 32, 2, 2))
    L.add_param("input_up_conv.bias", (32,))
    pre = "input_double_conv"
    L.add_param(pre + ".0.weight", (32, 32 + spec.in_channels, 3, 3))
    L.add_param(pre + ".0.bias", (32,))
    bn(pre + ".1", 32)
    L.add_param(pre + ".3.weight", (32, 32, 3, 3))
    L.add_param(pre + ".3.bias", (32,))
    bn(pre + ".4", 32)
    L.add_param("out_conv1x1.weight", (spec.num_classes, 32, 1, 1))
    L.add_param("out_conv1x1.bias", (spec.num_classes,))
    return L


def _bucket_backward(p: "_P", layout: ParamLayout, make_table, bucket_floats: int):
    """Split the backward program into segments after which a contiguous suffix bucket of the flat
    gradient buffer is final, and fold each bucket's conv-weight scratch (WGRAD_FINALIZE) right there.

    Backward visits layers in reverse, the flat layout is in registration order, so gradients become
    final roughly from the end of the buffer towards its start: bucket k = floats [lo_k, hi_k) with
    hi_0 = n_params, ready once every op writing into [lo_k, n_params) has been issued."""
    ops = p.bwd.ops
    last_write: dict[int, int] = {}   # param float offset -> index of the last op that writes its grad / scratch
    size_of = {off: (int(_numel(shape)) + 63) // 64 * 64 for off, shape in layout.params.values()}
    starts = sorted(size_of)
    import bisect

    for idx, (kind, fields) in enumerate(ops):
        for v in fields.values():
            if isinstance(v, TRef) and v.base in (D.BASE["GRADS"], D.BASE["WGS"]) and kind not in ("MEMSET",):
                fo = v.off // 4
                owner = starts[bisect.bisect_right(starts, fo) - 1]
                last_write[owner] = idx
    # suffix-ready index per parameter start (descending offsets)
    cuts = []   # (lo, hi, ready_op_index)
    hi = layout.n_params
    acc = 0
    ready = -1
    for off in reversed(starts):
        ready = max(ready, last_write.get(off, -1))
        acc += size_of[off]
        if acc >= bucket_floats or off == starts[0]:
            cuts.append((off, hi, ready))
            hi, acc = off, 0
    # make ready indices monotone (a later bucket can never be ready before an earlier one) and insert
    # the finalize ops back to front so indices stay valid
    entries = sorted(p.table, key=lambda r: r[0])
    inserts = []
    prev = -1
    for lo, hi_, r in cuts:
        r = max(r, prev)
        prev = r
        rows = [e for e in entries if lo <= e[0] < hi_]
        inserts.append((r, lo, hi_, rows))
    segments = []
    shift = 0
    begin = 0
    for r, lo, hi_, rows in inserts:
        pos = r + 1 + shift
        if rows:
            start = 0
            rel = []
            for off, M, C, T, _ in rows:
                rel.append([off, M, C, T, start])
                start += M * C * T
            ops.insert(pos, ("WGRAD_FINALIZE", dict(TABLE=make_table(rel, 5), WGS=TRef(D.BASE["WGS"], 0, (layout.n_params,)),
                                                    GRADS=TRef(D.BASE["GRADS"], 0, (layout.n_params,)), TOTAL=start,
                                                    N_ENTRIES=len(rel))))
            shift += 1
            pos += 1
        segments.append((begin, pos, lo, hi_))
        begin = pos
    if begin < len(ops):   # trailing ops (none write gradients): attach to the last segment
        a, b, lo, hi_ = segments[-1]
        segments[-1] = (a, len(ops), lo, hi_)
    return segments


def _numel(shape) -> int:
    n = 1
    for s_ in shape:
        n *= s_
    return n


_HBM_KINDS = ("DWCONV_FWD", "DWCONV_DGRAD", "DWCONV_WGRAD", "BN_BWD_REDUCE", "BN_BWD_APPLY", "SE_BWD_REDUCE", "SE_BN_SUMS", "BN_RESIDUAL")


def _defer_decoder_wgrads(ops: list, min_gflop: float = 4.0) -> list:
    """Re-order the backward program so that the large, matrix-core-bound weight gradients of the decoder are issued
    (on the executor's side stream) while the main stream works through the encoder's backward, which is mostly HBM-bound
    (depthwise, BatchNorm, SE): issued where the tape emits them they run next to the decoder's data-gradient convs and the
    two only share the matrix cores.  Nothing reads a weight gradient before the bucket's WGRAD_FINALIZE, which
    `_bucket_backward` places after the last writer in the NEW order; the operands (dY of the layer, forward activations)
    are never overwritten later in the backward (bump arena, in-place updates only touch gradient buffers of the layer
    being processed).  Deferred ops keep their relative order and are spread over the encoder section in proportion to its
    element traffic."""
    first = next((i for i, (k, _) in enumerate(ops) if k in ("DWCONV_DGRAD", "SE_BWD_REDUCE", "SE_BN_SUMS", "SE_FC_BWD")), None)
    if first is None:
        return ops

    def gflop(f):
        return 2.0e-9 * f["M"] * f["C"] * f["KH"] * f["KW"] * f["B"] * f["HO"] * f["WO"]

    moved = [(i, gflop(f)) for i, (k, f) in enumerate(ops[:first]) if k == "WGRAD" and gflop(f) >= min_gflop]
    if not moved:
        return ops
    take = {i for i, _ in moved}
    head = [op for i, op in enumerate(ops[:first]) if i not in take]
    tail = ops[first:]
    cost = [float(f["B"] * f["C"] * f.get("HW", f.get("H", 1) * f.get("W", 1))) if k in _HBM_KINDS else 0.0 for k, f in tail]
    total_cost, total_side = sum(cost) or 1.0, sum(g for _, g in moved)
    out, j, cum_c, cum_s = [], 0, 0.0, 0.0
    for (op, c) in zip(tail, cost):
        # issue the next deferred wgrad once the main stream's share of traffic catches up with the side work already issued
        while j < len(moved) and cum_c / total_cost >= cum_s / total_side:
            out.append(ops[moved[j][0]])
            cum_s += moved[j][1]
            j += 1
        out.append(op)
        cum_c += c
    out.extend(ops[i] for i, _ in moved[j:])
    return head + out


def finish_plan(p: "_P", layout: ParamLayout, training: bool, bucket_floats: int):
    """Common tail of every planner: emit the backward from the tape, zero the accumulators, cut the backward
    into bucketed segments (+ WGRAD_FINALIZE per bucket) and put the WEIGHT_PACK stages in front."""
    table = p.const_table

    q4 = {"base": 0, "packs_end": 0}

    def pack_op(prog: Program, rows, flags: int = 0):
        if not rows:
            return
        total = rows[-1][11] + rows[-1][10] * rows[-1][4] * rows[-1][9]
        extra = {"Q4_BASE": q4["base"]} if q4["base"] and any(r[8] & 2 for r in rows) else {}
        if flags:
            extra["_flags"] = flags
        prog.ops.insert(0, ("WEIGHT_PACK", dict(TABLE=table(rows, 12), SRC=TRef(D.BASE["PARAMS"], 0, (layout.n_params,)),
                                                DST=TRef(D.BASE["WPACK"], 0, (q4["packs_end"] // 4,)), TOTAL=total,
                                                N_ENTRIES=len(rows), **extra)))

    # f32 training plans: the BACKWARD's weight copies (transposed for the data gradients) are written during the FORWARD, on the
    # executor's side stream, which has nothing else to do there - 0.17 ms (U-Net b5) off the main queue, which IS the step
    # (profiles/r03_unet_streams.txt: busy 31.0 of 32.5 ms): +0.4 % in alternating runs (tools/exp_pack_side.sh; the ViT plans switch it off).  Legal because nothing changes the parameters between a forward and
    # its backward (the optimiser runs after it), the two programs' copies live in different parts of WPACK, and no forward stage
    # touches the backward's part; the side stream is forked from the caller's stream at the start of the run (so it sees the previous
    # step's Adam) and joined at its end.  bf16-mixed plans keep the pack in front of the backward (mark_bf16 looks for it there).
    bwd_pack_in_fwd = p.want_bwd and not getattr(p, "bf16", False) and getattr(p, "pack_side", True) and tune("S2K_PACK_SIDE", "1") != "0"

    def attach_q4():       # (once every CONV stage exists and before the pack tables are built)
        q4["packs_end"] = p.wpack.mark()
        if not getattr(p, "bf16", False):
            q4["base"] = mark_q4(p, p.bwd if p.want_bwd else None)

    def attach_splitk_scratch():
        """One shared scratch region (stages run in stream order) for the 1x1 convs with few output pixels and a long
        reduction: the kernel may then cut K into partial tiles (csrc/igemm.hip, splitk_reduce_kernel)."""
        want = []
        for prog in (p.fwd, p.bwd):
            for kind, f in prog.ops:
                if kind == "CONV" and f["KH"] == 1 and f["KW"] == 1 and f["MODE"] == D.MODE_CONV \
                        and f["B"] * f["HO"] * f["WO"] <= 8192 and f["C1"] + f["C2"] >= 256:
                    want.append((f, 8 * f["B"] * f["YC"] * f["HO"] * f["WO"]))
        if want:
            scratch = p.ws.alloc("splitk_scratch", (max(n for _, n in want),))
            for f, _ in want:
                f["SCRATCH"] = scratch

    fwd_aux_end = p.aux.mark()
    if fwd_aux_end:
        p.fwd.ops.insert(0, ("MEMSET", dict(DST=TRef(D.BASE["AUX"], 0, (fwd_aux_end,), "f32"), BYTES=fwd_aux_end)))

    bwd = None
    segments = []
    if p.want_bwd:
        for back in reversed(p.tape):
            back()
            p.marks.append(len(p.bwd.ops))
        bwd_aux_end = p.aux.mark()
        defer = getattr(p, "defer_wgrads", None)
        if defer is None:
            # f32: the decoder's large weight gradients (>= 8 GFLOP: 19 of 23 at the benchmark size) are issued during the encoder's
            # HBM-bound backward instead of beside the MFMA-bound decoder convs.  Re-measured in round 3 after the depthwise / bandwidth
            # kernels got faster (tools/exp_defer_ab.sh, alternating runs on one box): threshold 4 -> 8 GFLOP +0.5 % (the encoder has less
            # slack to hide them: deferred work finished ~0.13 ms after the main queue); bf16-mixed: no deferral at all +1.1 % (its decoder
            # convs are HBM-bound too, nothing is gained by moving MFMA work away from them), threshold 8 only +0.25 %.
            defer = tune("S2K_DEFER_WGRAD", "0" if getattr(p, "bf16", False) else "1") != "0"
        if defer:
            p.bwd.ops[:] = _defer_decoder_wgrads(p.bwd.ops, float(tune("S2K_DEFER_MIN_GFLOP", "8")))
        # zero the weight-gradient scratch of the convs that go through WGRAD_FINALIZE (3x3, 2x2-gather): one range from the
        # first to the last such weight (1x1 convs / Linears accumulate straight into the gradient buffer and need none)
        pre_ops = []
        if p.table:
            lo = min(r[0] for r in p.table)
            hi = max(r[0] + r[1] * r[2] * r[3] for r in p.table)
            pre_ops.append(("MEMSET", dict(DST=TRef(D.BASE["WGS"], lo * 4, (hi - lo,), "f32"), BYTES=(hi - lo) * 4)))
        if bwd_aux_end > fwd_aux_end:
            pre_ops.append(("MEMSET", dict(DST=TRef(D.BASE["AUX"], fwd_aux_end, (1,), "f32"),
                                           BYTES=bwd_aux_end - fwd_aux_end)))
        p.bwd.ops[0:0] = pre_ops
        segments = _bucket_backward(p, layout, table, bucket_floats)
        # weight / bias gradients feed nothing in the backward chain until the bucket's WGRAD_FINALIZE: the executor may run
        # them on its side stream, concurrently with the (mostly HBM-bound) BatchNorm / depthwise stages of the layers below
        side_max = float(tune("S2K_SIDE_MAX_GFLOP", "1e9")) * 1e9
        finalize_on_side = tune("S2K_FINALIZE_SIDE", "1") != "0"
        for kind, f in p.bwd.ops:
            if kind == "WGRAD" and 2.0 * f["M"] * f["C"] * f["KH"] * f["KW"] * f["B"] * f["HO"] * f["WO"] > side_max:
                continue      # two large MFMA-bound kernels side by side only fight for the same units
            if kind in ("WGRAD", "DWCONV_WGRAD", "CHANNEL_SUM", "SE_FC_WGRAD"):
                f["_flags"] = D.FLAG_SIDE
            elif kind == "WGRAD_FINALIZE":
                # the fold of a bucket's conv-weight scratch into the gradient buffer reads what the weight gradients wrote: on the
                # side stream it is ordered behind them by the stream itself and the main queue does not wait (a JOIN here stalled the
                # main queue ~0.17 ms per step in tape-order plans, where the decoder's bucket closes while its weight gradients are
                # still running); the executor joins at the end of every run, so a bucket is still final when its segment returns
                f["_flags"] = D.FLAG_SIDE if finalize_on_side else D.FLAG_JOIN
        side_stream_hazards(p.bwd)
        n_before = len(p.bwd.ops)
        attach_q4()
        if not bwd_pack_in_fwd:
            pack_op(p.bwd, p.pack_rows["bwd"])
        if len(p.bwd.ops) > n_before:   # WEIGHT_PACK went in front
            segments = [(a + 1 if a else 0, b + 1, lo, hi) for (a, b, lo, hi) in segments]
        bwd = p.bwd
    else:
        attach_q4()
    # (Also packing the forward's OWN later layers on the side stream, with a FLAG_JOIN on the first stage that reads one, was measured
    # and dropped: 32.04 vs 32.04 ms.)
    pack_op(p.fwd, p.pack_rows["fwd"])
    if bwd_pack_in_fwd:
        pack_op(p.fwd, p.pack_rows["bwd"], D.FLAG_SIDE)       # (in front of everything: forked before the forward's own pack is enqueued)
    attach_splitk_scratch()
    if getattr(p, "bf16", False):
        mark_bf16(p, bwd)
    return segments, bwd


def side_stream_hazards(prog: Program) -> int:
    """Write-after-read hazards between the two streams of the executor: a side-stream stage (weight / bias gradients) is only
    ordered AFTER the main-stream work issued before it (the fork event); nothing makes later main-stream work wait for it until
    the next FLAG_JOIN.  That is fine as long as nobody overwrites what it reads - true for the U-Net, whose gradients each have
    a buffer of their own, NOT for the transformer blocks, whose residual-stream gradient is accumulated IN PLACE (the LayerNorm
    backward adds into the buffer that the fc2 / proj weight and bias gradients are still reading on the side stream: a race
    that showed up as a run-order-dependent 1e-4 error of the MAE gradients).  Every main-stream stage that writes a byte range an
    outstanding side-stream stage reads gets FLAG_JOIN (the executor then waits for the side stream first).  Returns the number
    of joins added."""
    def ranges(f, names):
        out = []
        for k in names:
            v = f.get(k)
            if isinstance(v, TRef):
                out.append((v.base, v.off, v.off + v.nbytes))
        return out

    pending: list = []          # byte ranges read by side-stream stages since the last join
    added = 0
    for kind, f in prog.ops:
        flags = f.get("_flags", 0)
        t_names = D.OPS[kind][0]
        writes = D.WRITES.get(kind, tuple(t_names))
        if flags & D.FLAG_SIDE:
            pending.extend(ranges(f, [k for k in t_names if k not in writes]))
            continue
        if flags & D.FLAG_JOIN:
            pending.clear()
            continue
        if pending:
            for (b, lo, hi) in ranges(f, writes):
                if any(b == pb and lo < phi and plo < hi for (pb, plo, phi) in pending):
                    f["_flags"] = flags | D.FLAG_JOIN
                    pending.clear()
                    added += 1
                    break
    return added


def q4_conv_ok(f: dict) -> bool:
    """the stages csrc/conv_q4.hip takes: prologue-free 1x1 contractions over pixel quads (launch_conv_q4's shape list) where its
    measured routing rule sends them - the large maps' short reductions / thin layers and the deepest 8x8 layers"""
    if not (f["KH"] == 1 and f["KW"] == 1 and f["STRIDE"] == 1 and f["MODE"] == D.MODE_CONV and f["C2"] == 0 and f["PRO1"] == D.PRO_NONE
            and f.get("GATE1") is None and (f["H"] * f["W"]) % 4 == 0 and f["M"] >= 24 and not f.get("X1_BF16", 0)):
        return False
    n, m, c = f["B"] * f["H"] * f["W"], f["M"], f["C1"]
    return (n >= 32768 and m >= 40 and c < 256) or (n <= 8192 and m <= 512 and c >= 2048)


def mark_q4(p: "_P", bwd) -> int:
    """f32 plans, BEFORE the WEIGHT_PACK stages are emitted: the prologue-free 1x1 convs / Linears / data gradients that csrc/conv_q4.hip
    takes carry FLAG_Q4 and read, as WTB, a second f32 copy of their weight in that kernel's quad layout, which WEIGHT_PACK writes
    into a mirror region behind the packs (Q4_BASE; only the table rows marked here - flip bit 1 - are copied).  The arithmetic is
    unchanged (same f32 values, same k order): the emulator keeps reading WT.  Returns Q4_BASE (0: no such stage)."""
    if tune("S2K_Q4", "1") != "1":
        return 0
    base = (p.wpack.mark() + 255) // 256 * 256
    offs = set()
    for prog in (p.fwd, bwd):
        if prog is None:
            continue
        for kind, f in prog.ops:
            if kind == "CONV" and isinstance(f.get("WT"), TRef) and f["WT"].base == D.BASE["WPACK"] and f.get("WTB") is None and q4_conv_ok(f):
                f["_flags"] = f.get("_flags", 0) | D.FLAG_Q4
                f["WTB"] = TRef(D.BASE["WPACK"], base + f["WT"].off, f["WT"].shape, "f32", "q4:" + f["WT"].name)
                offs.add(f["WT"].off // 4)
    if not offs:
        return 0
    for rows in p.pack_rows.values():
        for row in rows:
            if row[1] in offs and row[4] == 1:
                row[8] |= 2
    p.wpack.top = 2 * base + 256
    return base


def mark_bf16(p: "_P", bwd) -> None:
    """bf16-MIXED plan (reported separately from the f32 parity path; the reference's own default is precision="bf16",
    configs/segmentation.py:146,153): every dense conv / Linear and every weight gradient MAY round its two MFMA operands to bf16
    (FLAG_BF16; f32 accumulation, f32 BatchNorm statistics, loss, master weights and optimiser).  WEIGHT_PACK writes a bf16 copy
    of every packed weight into a mirror region behind the f32 packs (BF16_BASE).  Flagged are exactly the shapes the bf16 kernels
    take (plan/bf16.py restates the shape lists of csrc/conv_bf16.hip, wgrad_bf16.hip); every other stage stays exact f32."""
    from . import bf16 as B16

    base = (p.wpack.mark() + 255) // 256 * 256
    for prog in (p.fwd, bwd):
        if prog is None:
            continue
        for kind, f in prog.ops:
            if kind == "WEIGHT_PACK":
                f["BF16_BASE"] = base
            elif kind == "CONV" and isinstance(f.get("WT"), TRef) and f["WT"].base == D.BASE["WPACK"] and B16.conv_ok(f):
                f["_flags"] = f.get("_flags", 0) | D.FLAG_BF16
                f["WTB"] = TRef(D.BASE["WPACK"], base + f["WT"].off // 2, f["WT"].shape, "i16", "bf16:" + f["WT"].name)
            elif kind == "WGRAD" and B16.wgrad_ok(f):
                f["_flags"] = f.get("_flags", 0) | D.FLAG_BF16
    p.wpack.top = base + (base + 1) // 2 + 256


def fold_bn_finalize(prog: Program) -> int:
    """Peephole over a forward program: a training BN_FINALIZE whose {scale, shift} are first read by a depthwise conv, an SE
    pool or a BN_RESIDUAL is folded INTO that stage (opdefs.FOLD_*: every workgroup derives the two numbers of its channels
    from the statistics itself, one of them publishes BNV and updates the running statistics) and its own 5-us launch
    disappears - 117 of the 126 per U-Net-b5 step, 0.9 ms of a 38.5 ms step.  Returns the number of folded stages."""
    if tune("S2K_FOLD_BN", "1") != "1":
        return 0
    ops = prog.ops
    folded = 0
    i = 0
    while i < len(ops):
        kind, f = ops[i]
        if kind != "BN_FINALIZE" or not f.get("TRAIN") or f.get("_flags"):
            i += 1
            continue
        bnv = f["BNV"]
        touched = {t.ref for k in ("BNV", "RM", "RV", "STATS") if isinstance(t := f.get(k), TRef)}
        target = None
        for j in range(i + 1, len(ops)):
            k2, f2 = ops[j]
            refs = {v.ref for v in f2.values() if isinstance(v, TRef)}
            if not (refs & touched):
                continue
            if (k2 in D.FOLD_KINDS and isinstance(f2.get("BNV"), TRef) and f2["BNV"].ref == bnv.ref and f2.get("FSTATS") is None
                    and f2.get("C") == f["C"] and f2.get("PRO", 1) and len(refs & touched) == 1 and not f2.get("_flags")):
                target = j
            break
        if target is None:
            i += 1
            continue
        ops[target][1].update(FSTATS=f["STATS"], FGAMMA=f["GAMMA"], FBETA=f["BETA"], FRM=f["RM"], FRV=f["RV"], FCOUNT=f["COUNT"],
                              FNREP=f["NREP"], FEPS=f["EPS"], FMOM=f["MOM"])
        del ops[i]
        folded += 1
    return folded


def fmap_block_indices(spec: UnetSpec, H: int, W: int) -> list[int]:
    """Blocks whose output EfficientNet.encode collects: the first block at each new spatial size, deepest first, without the
    size of the conv_head output (reference :255-260 with the (7, 7) literal generalised)."""
    h, w = same_pads(H, 3, 2)[0], same_pads(W, 3, 2)[0]
    sizes = []
    for b in spec.blocks:
        h, w = same_pads(h, b.kernel, b.stride)[0], same_pads(w, b.kernel, b.stride)[0]
        sizes.append((h, w))
    head = sizes[-1]
    out: list[int] = []
    seen: list[tuple] = []
    for i, sz in enumerate(sizes):
        if sz not in seen and sz != head:
            seen.append(sz)
            out.insert(0, i)
    return out


def emit_encoder(p: "_P", spec: UnetSpec, x_in: Act, pre: str = "encoder.", fmap_refs: dict | None = None):
    """EfficientNet.encode (reference :251-263): stem, MBConv blocks, conv_head.  Returns (head activation, feature maps deepest
    first, all block outputs).  fmap_refs: block index -> TRef where that block's output must be materialised."""
    eps, mom = spec.bn_eps, spec.bn_momentum
    cur = conv_bn(p, pre + "stem.0.weight", pre + "stem.1", [x_in], spec.stem_out, 3, 2, True, D.PRO_SILU, eps, mom)
    block_outs: list[Act] = []
    n = len(spec.blocks)
    for i, b in enumerate(spec.blocks):
        bp = f"{pre}blocks.{i}."
        xin = cur
        j = 0
        a = xin
        if b.expand != 1:
            a = conv_bn(p, bp + "stem.0.weight", bp + "stem.1", [xin], b.cexp, 1, 1, True, D.PRO_SILU, eps, mom)
            j = 3
        a = dwconv_bn(p, bp + f"stem.{j}.weight", bp + f"stem.{j + 1}", a, b.kernel, b.stride, eps, mom)
        a = squeeze_excite(p, bp + "squeeze_excitation", a, b.se)
        rate = spec.drop_connect_rate * (i / n) if spec.drop_connect_rate is not None else None
        cur = project_conv_bn_residual(p, i, bp + "final_layer.0.weight", bp + "final_layer.1", a, b.cout,
                                       xin if b.residual else None, rate, eps, mom,
                                       out_ref=(fmap_refs or {}).get(i))
        block_outs.append(cur)
    head_hw = (cur.H, cur.W)
    fmaps: list[Act] = []
    for a in block_outs:  # first block output at each new spatial size, deepest first (encode :259)
        if (a.H, a.W) not in [(f.H, f.W) for f in fmaps] and (a.H, a.W) != head_hw:
            fmaps.insert(0, a)
    if len(fmaps) != 4:
        raise ValueError(f"expected 4 skip feature maps, got {len(fmaps)} for input {x_in.H}x{x_in.W}")
    cur = conv_bn(p, pre + "conv_head.0.weight", pre + "conv_head.1", [cur], spec.head_out, 1, 1, True, D.PRO_SILU, eps, mom)
    return cur, fmaps, block_outs


def build_encoder_layout(spec: UnetSpec) -> ParamLayout:
    """Flat layout of a standalone EfficientNet (reference :179-244): the encoder entries of `build_layout` without the prefix."""
    full = build_layout(spec)
    L = ParamLayout()
    for name, (_, shape) in full.params.items():
        if name.startswith("encoder."):
            L.add_param(name[len("encoder."):], shape)
    for name, (_, shape) in full.bufs.items():
        if name.startswith("encoder."):
            L.add_buf(name[len("encoder."):], shape)
    L.nbt = [n[len("encoder."):] for n in full.nbt if n.startswith("encoder.")]
    return L


def plan_unet(spec: UnetSpec, B: int, H: int, W: int, training: bool, layout: ParamLayout | None = None,
              bucket_floats: int = 8 << 20, defer_wgrads: bool | None = None, want_bwd: bool | None = None,
              want_dx: bool = False, bf16: bool = False) -> UnetPlan:
    """defer_wgrads: None = the S2K_DEFER_WGRAD default (on); False keeps the decoder's weight gradients where the tape emits
    them, so that gradient buckets become final progressively (what the data-parallel reducer wants, see ddp.py).
    bf16: the bf16-mixed plan (mark_bf16)."""
    if H % 32 or W % 32:
        raise ValueError(f"EfficientnetUnet needs H, W multiples of 32, got {H}x{W}")
    layout = layout or build_layout(spec)
    p = _P(spec, layout, B, H, W, training, want_bwd)
    p.defer_wgrads = defer_wgrads
    p.bf16 = bool(bf16)
    eps, mom = spec.bn_eps, spec.bn_momentum
    # want_dx: the caller's input requires a gradient (torch semantics: x.requires_grad) — it lands in the DX base
    x_in = Act(TRef(D.BASE["X"], 0, (B, spec.in_channels, H, W), "f32", "x"), spec.in_channels, H, W,
               needs_grad=bool(want_dx), grad=TRef(D.BASE["DX"], 0, (B, spec.in_channels, H, W), "f32", "dx") if want_dx else None)

    # encoder --------------------------------------------------------------------------
    cur, fmaps, _ = emit_encoder(p, spec, x_in, "encoder.")

    # decoder --------------------------------------------------------------------------
    ups_out = [512, 256, 128, 64]
    cat = CAT_SIZES[spec.version]
    for i, fm in enumerate(fmaps):
        u = conv_transpose(p, f"up_convs.{i}.weight", f"up_convs.{i}.bias", cur, ups_out[i])
        if u.C + fm.C != cat[i] or (u.H, u.W) != (fm.H, fm.W):
            raise ValueError(f"decoder level {i}: cat {u.C}+{fm.C} != {cat[i]} or size mismatch")
        pre = f"double_convs.{i}"
        a = conv_bn(p, pre + ".0.weight", pre + ".1", [u, fm], ups_out[i], 3, 1, False, D.PRO_RELU, 1e-5, 0.1,
                    bias=pre + ".0.bias")
        cur = conv_bn(p, pre + ".3.weight", pre + ".4", [a], ups_out[i], 3, 1, False, D.PRO_RELU, 1e-5, 0.1,
                      bias=pre + ".3.bias")
    u = conv_transpose(p, "input_up_conv.weight", "input_up_conv.bias", cur, 32)
    pre = "input_double_conv"
    a = conv_bn(p, pre + ".0.weight", pre + ".1", [u, x_in], 32, 3, 1, False, D.PRO_RELU, 1e-5, 0.1,
                bias=pre + ".0.bias")
    cur = conv_bn(p, pre + ".3.weight", pre + ".4", [a], 32, 3, 1, False, D.PRO_RELU, 1e-5, 0.1,
                  bias=pre + ".3.bias")
    logits = out_conv(p, "out_conv1x1.weight", "out_conv1x1.bias", cur, spec.num_classes)

    segments, bwd = finish_plan(p, layout, training, bucket_floats)
    fold_bn_finalize(p.fwd)

    return UnetPlan(spec, B, H, W, training, p.fwd, bwd, p.ws.mark(), p.aux.mark(), p.blob, layout, len(spec.blocks),
                    segments if p.want_bwd else [], (B, spec.num_classes, H, W), p.tensors, p.wpack.mark())
