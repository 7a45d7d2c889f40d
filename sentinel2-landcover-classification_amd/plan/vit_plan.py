"""Planner for the Prithvi MAE-ViT path: MaskedAutoencoderViT (pre-training) and PrithviSegmentationNet
(ViT encoder + ConvTranspose neck + FCN head).  Emits forward / backward stage programs like unet_plan.

Data flow follows /root/reference/src/modules/prithvi.py (forward_encoder :285-305, forward_decoder :307-331,
forward_loss :333-350) and prithvi_segmentation.py (:23-72, :75-111, :156-162); the transformer block is timm's
(`Block`: pre-norm attention + MLP, restated in oracle/vit_block_ref.py — parity unpinned at that boundary).

Layout: ViT activations are FEATURE-MAJOR, [B][C][L] — tokens take the place of pixels — so that
  * every Linear is the 1x1 implicit-GEMM conv stage (tokens on the MFMA lanes, 128-B coalesced rows),
  * LayerNorm is the same channel-LayerNorm stage the neck's Norm2d needs,
  * q/k/v rows of one head are [hd][L] tiles that feed the MFMA operands of QK^T and PV without a transpose,
  * the neck consumes the encoder output as [B][768][14][14] directly.
The reference's token-major tensors (pred, latent) are produced by a TRANSPOSE_CL stage at the API boundary.

Token rows are padded to a multiple of 4 floats ([B][C][NS], NS = row_stride(N)): 50 -> 52 (MAE encoder), 197 -> 200, so
that every row is 16-byte aligned and the conv / attention stagers use their vector loads.  The padding columns are
ordinary pixels to the conv / LayerNorm / activation stages (finite values, 1.5-4 % extra work); attention ignores
them as keys and writes zeros for them; every gradient tensor is exactly zero there (the loss, token-scatter and
attention backward stages write zeros, all other backward stages map zero columns to zero columns), so weight,
bias and LayerNorm-parameter gradients summed over NS columns equal the sums over the N real tokens.
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field

from . import opdefs as D
from .program import Program, TRef
from .unet_plan import Act, ParamLayout, _P, _conv_dgrad_wgrad, _numel, conv_bn, conv_transpose, finish_plan


@dataclass
class MaeSpec:
    img_size: int = 224
    patch_size: int = 16
    num_frames: int = 1
    tubelet_size: int = 1
    in_chans: int = 6
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    norm_pix_loss: bool = False
    decoder: bool = True

    @property
    def grid(self):
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet_size, g, g)

    @property
    def num_patches(self) -> int:
        t, h, w = self.grid
        return t * h * w

    @property
    def patch_dim(self) -> int:
        return self.tubelet_size * self.patch_size * self.patch_size * self.in_chans


@dataclass
class SegSpec:
    mae: MaeSpec
    num_classes: int
    fcn_out_channels: int
    fcn_num_convs: int
    fcn_dropout: float
    frozen_backbone: bool

    @property
    def embed(self) -> int:
        return self.mae.embed_dim * self.mae.num_frames


@dataclass
class VitPlan:
    spec: object
    B: int
    training: bool
    fwd: Program
    bwd: Program | None
    ws_bytes: int
    aux_bytes: int
    const_table: list
    layout: ParamLayout
    bwd_param_marks: list
    outputs: dict            # name -> TRef in the OUT base
    out_bytes: int
    noise: dict              # name -> TRef in the NOISE base (caller-supplied uniforms)
    noise_bytes: int
    dout_shape: tuple        # what the caller provides in DOUT (upstream gradient)
    tensors: dict
    wpack_bytes: int = 0
    trainable_lo: int = 0    # flat floats [trainable_lo, n_params) receive gradients (frozen backbone sits in front)
    x_shape: tuple = ()
    douts: dict | None = None   # several differentiable outputs: name -> TRef in a packed DOUT buffer (None: `dout_shape` only)
    dout_bytes: int = 0
    want_dx: bool = False       # the backward program also writes the gradient w.r.t. the images into the DX base
    # name of a differentiable output -> [i0, i1): the backward stages that only move ITS upstream gradient into place; when the
    # caller has no gradient for that output (the trainer differentiates the loss only) the executor skips them and passes a
    # DOUT buffer without that region
    optional_dout_ops: dict | None = None


# ---- layouts (reference registration order) ---------------------------------------------
def _block_params(L: ParamLayout, prefix: str, dim: int, hidden: int):
    L.add_param(prefix + ".norm1.weight", (dim,))
    L.add_param(prefix + ".norm1.bias", (dim,))
    L.add_param(prefix + ".attn.qkv.weight", (3 * dim, dim))
    L.add_param(prefix + ".attn.qkv.bias", (3 * dim,))
    L.add_param(prefix + ".attn.proj.weight", (dim, dim))
    L.add_param(prefix + ".attn.proj.bias", (dim,))
    L.add_param(prefix + ".norm2.weight", (dim,))
    L.add_param(prefix + ".norm2.bias", (dim,))
    L.add_param(prefix + ".mlp.fc1.weight", (hidden, dim))
    L.add_param(prefix + ".mlp.fc1.bias", (hidden,))
    L.add_param(prefix + ".mlp.fc2.weight", (dim, hidden))
    L.add_param(prefix + ".mlp.fc2.bias", (dim,))


def mae_layout(s: MaeSpec, prefix: str = "", L: ParamLayout | None = None) -> ParamLayout:
    """The fixed sin-cos tables are requires_grad=False nn.Parameters in the reference (prithvi.py:155-157,173-175): they keep
    their place in the flat parameter buffer (same state_dict order) and are listed in `L.frozen`, which the optimiser and
    the gradient publication skip."""
    L = L or ParamLayout()
    if not hasattr(L, "frozen"):
        L.frozen = set()
    Lp, Dm, Dd = s.num_patches, s.embed_dim, s.decoder_embed_dim
    L.add_param(prefix + "cls_token", (1, 1, Dm))
    L.add_param(prefix + "pos_embed", (1, Lp + 1, Dm))
    L.frozen.add(prefix + "pos_embed")
    if s.decoder:
        L.add_param(prefix + "mask_token", (1, 1, Dd))
    L.add_param(prefix + "decoder_pos_embed", (1, Lp + 1, Dd))
    L.frozen.add(prefix + "decoder_pos_embed")
    L.add_param(prefix + "patch_embed.proj.weight", (Dm, s.in_chans, s.tubelet_size, s.patch_size, s.patch_size))
    L.add_param(prefix + "patch_embed.proj.bias", (Dm,))
    for i in range(s.depth):
        _block_params(L, f"{prefix}blocks.{i}", Dm, int(Dm * s.mlp_ratio))
    L.add_param(prefix + "norm.weight", (Dm,))
    L.add_param(prefix + "norm.bias", (Dm,))
    if s.decoder:
        L.add_param(prefix + "decoder_embed.weight", (Dd, Dm))
        L.add_param(prefix + "decoder_embed.bias", (Dd,))
        for i in range(s.decoder_depth):
            _block_params(L, f"{prefix}decoder_blocks.{i}", Dd, int(Dd * s.mlp_ratio))
        L.add_param(prefix + "decoder_norm.weight", (Dd,))
        L.add_param(prefix + "decoder_norm.bias", (Dd,))
        L.add_param(prefix + "decoder_pred.weight", (s.patch_dim, Dd))
        L.add_param(prefix + "decoder_pred.bias", (s.patch_dim,))
    return L


def seg_layout(s: SegSpec) -> ParamLayout:
    L = mae_layout(s.mae, "backbone.")
    E = s.embed
    nk = "neck.feature_pyramid_net."
    for i in (0, 3, 4, 7):
        L.add_param(f"{nk}{i}.weight", (E, E, 2, 2))
        L.add_param(f"{nk}{i}.bias", (E,))
        if i in (0, 4):
            L.add_param(f"{nk}{i + 1}.ln.weight", (E,))
            L.add_param(f"{nk}{i + 1}.ln.bias", (E,))
    cin, idx = E, 0
    for _ in range(s.fcn_num_convs):
        L.add_param(f"head.net.{idx}.weight", (s.fcn_out_channels, cin, 3, 3))
        L.add_param(f"head.net.{idx}.bias", (s.fcn_out_channels,))
        bn = f"head.net.{idx + 1}"
        L.add_param(bn + ".weight", (s.fcn_out_channels,))
        L.add_param(bn + ".bias", (s.fcn_out_channels,))
        L.add_buf(bn + ".running_mean", (s.fcn_out_channels,))
        L.add_buf(bn + ".running_var", (s.fcn_out_channels,))
        L.nbt.append(bn + ".num_batches_tracked")
        cin = s.fcn_out_channels
        idx += 3
    idx += 1
    L.add_param(f"head.net.{idx}.weight", (s.num_classes, cin, 1, 1))
    L.add_param(f"head.net.{idx}.bias", (s.num_classes,))
    return L


# fc2's data gradient multiplied by gelu'(fc1 output) in the CONV epilogue (FLAG_RES_GELU_GRAD) instead of a separate ACT_BWD pass.
# Implemented and tested (op level: tests/test_ops_gpu.py::test_conv1x1_times_gelu_grad_of_res; plan level: test_vit_plan_cpu.py), OFF by
# default: the MAE backward is bound by MFMA capacity - the side stream's weight gradients fill the chip while the main stream runs the
# HBM-bound ACT_BWD, so removing that pass frees nothing, and the erf in the epilogue lengthens the MFMA kernel instead (alternating
# runs on one box, tools/exp_gelu_fuse.sh: 1471.0 samples/s fused vs 1473.0 separate)
FUSE_GELU_GRAD = os.environ.get("S2LC_FUSE_GELU_GRAD", "0") == "1"
_ROW_ALIGN = 4      # floats; 8 while a bf16-mixed plan is being built (the bf16 weight-gradient kernel contracts pixel OCTETS)


def row_stride(n: int) -> int:
    return (n + _ROW_ALIGN - 1) // _ROW_ALIGN * _ROW_ALIGN


class _row_align:
    def __init__(self, floats: int):
        self.floats = floats

    def __enter__(self):
        global _ROW_ALIGN
        self.prev, _ROW_ALIGN = _ROW_ALIGN, self.floats

    def __exit__(self, *exc):
        global _ROW_ALIGN
        _ROW_ALIGN = self.prev


# ---- building blocks ----------------------------------------------------------------------------------
class _V:
    """ViT emission helpers on top of the shared planner state."""

    def __init__(self, p: _P, trainable):
        self.p = p
        p.pack_side = False            # the backward's weight copies stay in front of the backward: written during the forward on the side
                                       # stream they cost the MAE 0.15 % (2 x 344 MB of HBM traffic beside MFMA-bound Linears that have no
                                       # slack; the U-Net gains 0.4 %: tools/exp_pack_side.sh)
        self.trainable = trainable     # name -> bool (frozen backbone: no parameter gradients, no wgrad stages)
        self._ident: dict[int, TRef] = {}

    def ident_bnv(self, C: int) -> TRef:
        """{scale 1, shift 0, mean 0, invstd 1}[C]: lets a conv apply a bare activation (GELU) as its load prologue."""
        if C not in self._ident:
            self._ident[C] = self.p.const_floats([1.0] * C + [0.0] * C + [0.0] * C + [1.0] * C, (4, C))
        return self._ident[C]

    # Linear over feature-major tokens = 1x1 conv stage.  y[B][M][N] = W[M][K] x + b (+ res)
    def linear_fwd(self, wname: str, bname: str | None, x: TRef, K: int, M: int, N: int, pro: int = D.PRO_NONE,
                   res: TRef | None = None, out: TRef | None = None) -> TRef:
        p, B = self.p, self.p.B
        y = out if out is not None else p.alloc("y:" + wname, (B, M, N))
        wp, MP = p.pack_weight("fwd", wname, M, K, 1, K, 1, 1, 0)
        p.fwd.add("CONV", X1=x, BNV1=self.ident_bnv(K) if pro != D.PRO_NONE else None, GATE1=None, X2=None, BNV2=None, WT=wp,
                  BIAS=p.param(bname) if bname else None, Y=y, STATS=None, RES=res, B=B, C1=K, C2=0, H=1, W=N, M=M, KH=1, KW=1,
                  STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=N, PRO1=pro, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0,
                  BETA=0, YC=M, NREP=1)
        return y

    def linear_bwd(self, wname: str, bname: str | None, x: TRef, dy: TRef, K: int, M: int, N: int, pro: int = D.PRO_NONE,
                   dx: TRef | None = None, dx_beta: int = 0, bias_sum: bool = True, dx_gelu_of: TRef | None = None) -> None:
        """weight / bias gradients (when trainable) and, if `dx` is given, the input gradient w.r.t. pro(x).
        bias_sum=False: the bias gradient (the row sums of dy) is produced by the stage that wrote dy (CHAN_LN_BWD's DSUM).
        dx_gelu_of: x = gelu(that tensor) - the data-gradient stage multiplies by gelu'(it) in its epilogue (FLAG_RES_GELU_GRAD), so dx
        is the gradient w.r.t. the tensor BEFORE the GELU (no separate ACT_BWD pass over dx)."""
        p, B = self.p, self.p.B
        if self.trainable(wname):
            # [M][K] scratch layout = the Linear weight's layout: accumulate straight into the gradient buffer (no finalize pass)
            p.bwd.add("WGRAD", P=dy, BNVP=None, GATEP=None, Q=x, BNVQ=self.ident_bnv(K) if pro != D.PRO_NONE else None, GATEQ=None,
                      WGS=p.pgrad(wname), B=B, M=M, C=K, CTOT=K, H=1, W=N, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=N,
                      PROP=D.PRO_NONE, PROQ=pro, MODE=D.MODE_CONV)
            if bname and bias_sum:
                p.bwd.add("CHANNEL_SUM", G=dy, OUT=p.pgrad(bname), B=B, C=M, HW=N)
        if dx is not None:
            wp, MP = p.pack_weight("bwd", wname, K, M, 1, 1, K, 1, 0)
            if dx_gelu_of is not None and dx_beta:
                raise ValueError("a data gradient through a GELU is written, not accumulated")
            p.bwd.add("CONV", X1=dy, BNV1=None, GATE1=None, X2=None, BNV2=None, WT=wp, BIAS=None, Y=dx, STATS=None, RES=dx_gelu_of,
                      B=B, C1=M, C2=0, H=1, W=N, M=K, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=N, PRO1=D.PRO_NONE, PRO2=0,
                      MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=dx_beta, YC=K, NREP=1,
                      **({"_flags": D.FLAG_RES_GELU_GRAD} if dx_gelu_of is not None else {}))

    def ln_fwd(self, prefix: str, x: TRef, C: int, N: int, eps: float):
        p, B = self.p, self.p.B
        y = p.alloc("ln:" + prefix, (B, C, N))
        mr = p.alloc("mr:" + prefix, (B, N, 2))
        p.fwd.add("CHAN_LN_FWD", X=x, GAMMA=p.param(prefix + ".weight"), BETA=p.param(prefix + ".bias"), Y=y, MR=mr,
                  B=B, C=C, HW=N, EPS=eps)
        return y, mr

    def ln_bwd(self, prefix: str, dy: TRef, x: TRef, mr: TRef, dx: TRef, C: int, N: int, accum: int, dxin: TRef | None = None,
               dsum: TRef | None = None):
        """dxin (with accum): DX = DXIN + ... (out of place); dsum: += the row sums of the new DX (a Linear's bias gradient)."""
        p, B = self.p, self.p.B
        tr = self.trainable(prefix + ".weight")
        p.bwd.add("CHAN_LN_BWD", DY=dy, X=x, MR=mr, GAMMA=p.param(prefix + ".weight"), DX=dx,
                  DGAMMA=p.pgrad(prefix + ".weight") if tr else None, DBETA=p.pgrad(prefix + ".bias") if tr else None,
                  DXIN=dxin, DSUM=dsum, B=B, C=C, HW=N, ACCUM=accum)

    def bias_grad(self, bname: str) -> TRef | None:
        return self.p.pgrad(bname) if self.trainable(bname) else None

    # timm Block on [B][Dm][N] ------------------------------------------------------------------------
    def block_fwd(self, prefix: str, x: TRef, Dm: int, heads: int, hidden: int, L: int) -> tuple[TRef, dict]:
        """L real tokens in rows of N = row_stride(L) floats."""
        p, B = self.p, self.p.B
        hd = Dm // heads
        N = row_stride(L)
        h1, mr1 = self.ln_fwd(prefix + ".norm1", x, Dm, N, 1e-5)
        qkv = self.linear_fwd(prefix + ".attn.qkv.weight", prefix + ".attn.qkv.bias", h1, Dm, 3 * Dm, N)
        o = p.alloc("attn:" + prefix, (B, Dm, N))
        lse = p.alloc("lse:" + prefix, (B, heads, N))
        p.fwd.add("ATTN_FWD", QKV=qkv, O=o, LSE=lse, B=B, HEADS=heads, HD=hd, L=L, LS=N, SCALE=float(hd) ** -0.5)
        xm = self.linear_fwd(prefix + ".attn.proj.weight", prefix + ".attn.proj.bias", o, Dm, Dm, N, res=x)
        h2, mr2 = self.ln_fwd(prefix + ".norm2", xm, Dm, N, 1e-5)
        f1 = self.linear_fwd(prefix + ".mlp.fc1.weight", prefix + ".mlp.fc1.bias", h2, Dm, hidden, N)
        # GELU is materialised once (ACT_FWD): as a conv / wgrad prologue its erf polynomial would be re-evaluated for every
        # output-channel tile of fc2 and of fc2's weight gradient, and vector instructions are not free beside the f32 MFMA
        a1 = p.alloc("gelu:" + prefix, (B, hidden, N))
        p.fwd.add("ACT_FWD", X=f1, Y=a1, COUNT=B * hidden * N, ACT=D.ACT_GELU)
        xo = self.linear_fwd(prefix + ".mlp.fc2.weight", prefix + ".mlp.fc2.bias", a1, hidden, Dm, N, res=xm)
        return xo, dict(prefix=prefix, x=x, h1=h1, mr1=mr1, qkv=qkv, o=o, lse=lse, xm=xm, h2=h2, mr2=mr2, f1=f1, a1=a1, Dm=Dm, heads=heads,
                        hidden=hidden, N=N, L=L)

    def block_bwd(self, r: dict, g: TRef, below_fc2_bias: TRef | None = None) -> TRef:
        """g = gradient of the block output, [B][Dm][N]; returns the gradient of the block INPUT.  The residual stream's gradient
        gets a FRESH buffer at each of its two updates (CHAN_LN_BWD with DXIN): the fc2 / proj weight-gradient stages, which read
        it on the side stream, are never overwritten - an in-place accumulation raced with them.  The bias gradients of proj and
        of the fc2 of the block BELOW (`below_fc2_bias`) are the row sums of those new buffers and come out of the same
        CHAN_LN_BWD stages (DSUM); this block's own fc2 bias gradient was produced by whoever wrote `g`."""
        p, B = self.p, self.p.B
        pre, Dm, heads, hidden, N = r["prefix"], r["Dm"], r["heads"], r["hidden"], r["N"]
        g_f1 = p.alloc("g:f1:" + pre, (B, hidden, N))
        if getattr(p, "bf16", False) or not FUSE_GELU_GRAD:       # the bf16 kernels' epilogues add RES only: GELU' as its own pass
            self.linear_bwd(pre + ".mlp.fc2.weight", pre + ".mlp.fc2.bias", r["a1"], g, hidden, Dm, N, dx=g_f1, bias_sum=False)
            p.bwd.add("ACT_BWD", G=g_f1, X=r["f1"], COUNT=B * hidden * N, ACT=D.ACT_GELU)
        else:            # fc2's data gradient comes out already multiplied by gelu'(fc1 output): one pass over [B][hidden][N] less
            self.linear_bwd(pre + ".mlp.fc2.weight", pre + ".mlp.fc2.bias", r["a1"], g, hidden, Dm, N, dx=g_f1, bias_sum=False, dx_gelu_of=r["f1"])
        g_h = p.alloc("g:h:" + pre, (B, Dm, N))       # shared scratch for both LayerNorm output gradients
        self.linear_bwd(pre + ".mlp.fc1.weight", pre + ".mlp.fc1.bias", r["h2"], g_f1, Dm, hidden, N, dx=g_h)
        g_mid = p.alloc("g:mid:" + pre, (B, Dm, N))   # d loss / d (x + attention branch)
        self.ln_bwd(pre + ".norm2", g_h, r["xm"], r["mr2"], g_mid, Dm, N, accum=1, dxin=g, dsum=self.bias_grad(pre + ".attn.proj.bias"))
        g_o = p.alloc("g:o:" + pre, (B, Dm, N))
        self.linear_bwd(pre + ".attn.proj.weight", pre + ".attn.proj.bias", r["o"], g_mid, Dm, Dm, N, dx=g_o, bias_sum=False)
        g_qkv = p.alloc("g:qkv:" + pre, (B, 3 * Dm, N))
        p.bwd.add("ATTN_BWD", QKV=r["qkv"], DO=g_o, DQKV=g_qkv, O=r["o"], LSE=r["lse"], DELTA=p.alloc("delta:" + pre, (B, heads, N)),
                  B=B, HEADS=heads, HD=Dm // heads, L=r["L"], LS=N, SCALE=float(Dm // heads) ** -0.5)
        self.linear_bwd(pre + ".attn.qkv.weight", pre + ".attn.qkv.bias", r["h1"], g_qkv, Dm, 3 * Dm, N, dx=g_h)
        g_in = p.alloc("g:in:" + pre, (B, Dm, N))
        self.ln_bwd(pre + ".norm1", g_h, r["x"], r["mr1"], g_in, Dm, N, accum=1, dxin=g_mid, dsum=below_fc2_bias)
        return g_in

    def blocks_bwd(self, recs: list, g: TRef) -> TRef:
        """Backward through a stack of blocks (last first).  `g` must have been written by a CHAN_LN_BWD whose DSUM was
        `top_fc2_bias(recs)` (the final norm's backward)."""
        for i in range(len(recs) - 1, -1, -1):
            below = self.bias_grad(recs[i - 1]["prefix"] + ".mlp.fc2.bias") if i > 0 else None
            g = self.block_bwd(recs[i], g, below)
        return g

    def top_fc2_bias(self, recs: list) -> TRef | None:
        return self.bias_grad(recs[-1]["prefix"] + ".mlp.fc2.bias") if recs else None


def _encoder(v: _V, s: MaeSpec, prefix: str, x_img: TRef, noise: TRef, keep: int, outs: dict, need_input_grads: bool):
    """forward_encoder (prithvi.py:285-305).  Returns (latent [B][Dm][row_stride(N)], N, record for the backward)."""
    p, B = v.p, v.p.B
    C, T, Hh, P, tub = s.in_chans, s.num_frames, s.img_size, s.patch_size, s.tubelet_size
    Lp, Dm = s.num_patches, s.embed_dim
    Kp = C * tub * P * P
    cols = p.alloc("cols:" + prefix, (B, Kp, Lp))
    p.fwd.add("PATCHIFY", X=x_img, OUT=cols, B=B, C=C, T=T, H=Hh, W=Hh, P=P, TUB=tub)
    pe = v.linear_fwd(prefix + "patch_embed.proj.weight", prefix + "patch_embed.proj.bias", cols, Kp, Dm, Lp)
    enc_idx = p.alloc("enc_idx:" + prefix, (B, 1 + keep), "i32")
    dec_idx = p.alloc("dec_idx:" + prefix, (B, 1 + Lp), "i32")
    p.fwd.add("MAE_MASK_INDEX", NOISE=noise, IDS_RESTORE=outs["ids_restore"], MASK=outs["mask"], ENC_IDX=enc_idx, DEC_IDX=dec_idx,
              B=B, L=Lp, KEEP=keep)
    N = 1 + keep
    NS = row_stride(N)
    x0 = p.alloc("x0:" + prefix, (B, Dm, NS))
    p.fwd.add("TOKEN_GATHER", IN=pe, IDX=enc_idx, FILL=p.param(prefix + "cls_token"), POS=p.param(prefix + "pos_embed"), OUT=x0,
              B=B, C=Dm, LIN=Lp, LOUT=N, POS_BY_SRC=1, POS_OFF=1, LIN_S=Lp, LOUT_S=NS)
    recs = []
    x = x0
    for i in range(s.depth):
        x, r = v.block_fwd(f"{prefix}blocks.{i}", x, Dm, s.num_heads, int(Dm * s.mlp_ratio), N)
        recs.append(r)
    latent, mr = v.ln_fwd(prefix + "norm", x, Dm, NS, 1e-5)
    rec = dict(prefix=prefix, cols=cols, pe=pe, enc_idx=enc_idx, dec_idx=dec_idx, x_last=x, mr=mr, blocks=recs, N=N, NS=NS, Kp=Kp, Lp=Lp,
               Dm=Dm)
    return latent, N, rec


def _encoder_bwd(v: _V, s: MaeSpec, rec: dict, g_latent: TRef, dx: TRef | None = None, dx_accum: bool = False):
    """g_latent: gradient of the normalised encoder output [B][Dm][NS] (consumed; zero in the padding columns).
    dx: where to put the gradient w.r.t. the images (DX base) when the caller's input requires one."""
    p, B = v.p, v.p.B
    prefix, N, NS, Dm, Lp, Kp = rec["prefix"], rec["N"], rec["NS"], rec["Dm"], rec["Lp"], rec["Kp"]
    g = p.alloc("g:enc:" + prefix, (B, Dm, NS))
    v.ln_bwd(prefix + "norm", g_latent, rec["x_last"], rec["mr"], g, Dm, NS, accum=0, dsum=v.top_fc2_bias(rec["blocks"]))
    g = v.blocks_bwd(rec["blocks"], g)
    g_pe = p.alloc("g:pe:" + prefix, (B, Dm, Lp))
    p.bwd.add("TOKEN_SCATTER", DOUT=g, IDX=rec["enc_idx"], DIN=g_pe, DFILL=p.pgrad(prefix + "cls_token") if v.trainable(prefix + "cls_token") else None,
              B=B, C=Dm, LIN=Lp, LOUT=N,
              LIN_S=Lp, LOUT_S=NS)
    g_cols = p.alloc("g:cols:" + prefix, (B, Kp, Lp)) if dx is not None else None
    v.linear_bwd(prefix + "patch_embed.proj.weight", prefix + "patch_embed.proj.bias", rec["cols"], g_pe, Kp, Dm, Lp, dx=g_cols)
    if dx is not None:     # patches do not overlap: the column gradient is a permutation of the image gradient
        p.bwd.add("PATCHIFY", X=dx, OUT=g_cols, B=B, C=s.in_chans, T=s.num_frames, H=s.img_size, W=s.img_size, P=s.patch_size,
                  TUB=s.tubelet_size, INVERSE=3 if dx_accum else 1)


def _out(outs: dict, cursor: list, name: str, shape: tuple, dtype: str = "f32") -> TRef:
    isz = {"f32": 4, "i64": 8, "i32": 4}[dtype]
    t = TRef(D.BASE["OUT"], cursor[0], shape, dtype, name)
    outs[name] = t
    cursor[0] += (_numel(shape) * isz + 255) // 256 * 256
    return t


def plan_mae(s: MaeSpec, B: int, mask_ratio: float, training: bool, layout: ParamLayout | None = None,
             bucket_floats: int = 8 << 20, want_dx: bool = False, bf16: bool = False) -> VitPlan:
    """MaskedAutoencoderViT.forward(imgs, mask_ratio) -> (loss, pred, mask) (+ latent, ids_restore for forward_encoder).
    bf16: the bf16-mixed plan (unet_plan.mark_bf16: Linears and their weight gradients on bf16 MFMA operands; attention,
    LayerNorm, GELU, loss and the optimiser stay f32); token rows are padded to 8 floats instead of 4."""
    with _row_align(8 if bf16 else 4):
        return _plan_mae(s, B, mask_ratio, training, layout, bucket_floats, want_dx, bf16)


def _plan_mae(s: MaeSpec, B: int, mask_ratio: float, training: bool, layout, bucket_floats: int, want_dx: bool, bf16: bool) -> VitPlan:
    assert s.decoder
    layout = layout or mae_layout(s)
    p = _P(s, layout, B, s.img_size, s.img_size, training)
    p.bf16 = bool(bf16)
    v = _V(p, lambda name: True)
    Lp, Dm, Dd, PD = s.num_patches, s.embed_dim, s.decoder_embed_dim, s.patch_dim
    keep = int(Lp * (1 - mask_ratio))
    outs: dict = {}
    cur = [0]
    x_shape = (B, s.in_chans, s.num_frames, s.img_size, s.img_size)
    x_img = TRef(D.BASE["X"], 0, x_shape, "f32", "imgs")
    noise = TRef(D.BASE["NOISE"], 0, (B, Lp), "f32", "noise")
    _out(outs, cur, "loss", (1,))
    _out(outs, cur, "pred", (B, Lp, PD))
    _out(outs, cur, "mask", (B, Lp))
    _out(outs, cur, "ids_restore", (B, Lp), "i64")
    latent, N, erec = _encoder(v, s, "", x_img, noise, keep, outs, True)
    _out(outs, cur, "latent", (B, N, Dm))
    NS = erec["NS"]
    p.fwd.add("TRANSPOSE_CL", X=latent, Y=outs["latent"], B=B, C=Dm, L=NS, L_OFF=0, LOUT=N)
    # decoder (:307-331)
    dx = v.linear_fwd("decoder_embed.weight", "decoder_embed.bias", latent, Dm, Dd, NS)
    ND = Lp + 1
    NDS = row_stride(ND)
    y0 = p.alloc("y0", (B, Dd, NDS))
    p.fwd.add("TOKEN_GATHER", IN=dx, IDX=erec["dec_idx"], FILL=p.param("mask_token"), POS=p.param("decoder_pos_embed"), OUT=y0,
              B=B, C=Dd, LIN=N, LOUT=ND, POS_BY_SRC=0, POS_OFF=0, LIN_S=NS, LOUT_S=NDS)
    drecs = []
    y = y0
    for i in range(s.decoder_depth):
        y, r = v.block_fwd(f"decoder_blocks.{i}", y, Dd, s.decoder_num_heads, int(Dd * s.mlp_ratio), ND)
        drecs.append(r)
    yn, mrd = v.ln_fwd("decoder_norm", y, Dd, NDS, 1e-5)
    pred_fm = v.linear_fwd("decoder_pred.weight", "decoder_pred.bias", yn, Dd, PD, NDS)
    p.fwd.add("TRANSPOSE_CL", X=pred_fm, Y=outs["pred"], B=B, C=PD, L=NDS, L_OFF=1, LOUT=Lp)
    acc = p.aux.alloc("mae_acc", (2,), "f64")
    geo = dict(B=B, C=s.in_chans, T=s.num_frames, H=s.img_size, W=s.img_size, P=s.patch_size, TUB=s.tubelet_size, LP=NDS, L_OFF=1,
               NORM_PIX=int(s.norm_pix_loss))
    p.fwd.add("MAE_LOSS_FWD", PRED=pred_fm, IMGS=x_img, MASK=outs["mask"], LOSS=outs["loss"], ACC=acc, **geo)

    # upstream gradients: the loss (what the reference's trainer uses) and `pred` (torch lets a caller differentiate through
    # it too); both arrive in one packed DOUT buffer, an unused one as zeros
    d_loss = TRef(D.BASE["DOUT"], 0, (1,), "f32", "dloss")
    d_pred = TRef(D.BASE["DOUT"], 256, (B, Lp, PD), "f32", "dpred")
    x_dx = TRef(D.BASE["DX"], 0, x_shape, "f32", "dx") if want_dx else None

    def backward():
        gout = d_loss
        g_pred = p.alloc("g:pred", (B, PD, NDS))
        p.bwd.add("MAE_LOSS_BWD", PRED=pred_fm, IMGS=x_img, MASK=outs["mask"], ACC=acc, GOUT=gout, DPRED=g_pred, **geo)
        if x_dx is not None:
            # the loss also depends on the images through its TARGET (patchify(imgs), prithvi.py:340): d loss / d target = -d loss /
            # d pred; written first (it covers every pixel), the encoder's part is added at the end of the backward
            # (norm_pix_loss: through the per-patch standardisation as well, PATCHIFY INVERSE 4)
            p.bwd.add("PATCHIFY", X=x_dx, OUT=g_pred, IMGS=x_img if s.norm_pix_loss else None, B=B, C=s.in_chans, T=s.num_frames,
                      H=s.img_size, W=s.img_size, P=s.patch_size, TUB=s.tubelet_size, INVERSE=4 if s.norm_pix_loss else 2, ORDER=1,
                      LS=NDS, L_OFF=1)
        g_user = p.alloc("g:pred_user", (B, PD, NDS))
        p.bwd.add("TRANSPOSE_CL", X=d_pred, Y=g_user, B=B, C=Lp, L=PD, L_OFF=0, LOUT=PD, YS=NDS, Y_OFF=1)
        p.bwd.add("AXPY", X=g_user, Y=g_pred, COUNT=B * PD * NDS)
        g_yn = p.alloc("g:yn", (B, Dd, NDS))
        v.linear_bwd("decoder_pred.weight", "decoder_pred.bias", yn, g_pred, Dd, PD, NDS, dx=g_yn)
        g = p.alloc("g:dec", (B, Dd, NDS))
        v.ln_bwd("decoder_norm", g_yn, y, mrd, g, Dd, NDS, accum=0, dsum=v.top_fc2_bias(drecs))
        g = v.blocks_bwd(drecs, g)
        g_dx = p.alloc("g:dx", (B, Dd, NS))
        p.bwd.add("TOKEN_SCATTER", DOUT=g, IDX=erec["dec_idx"], DIN=g_dx, DFILL=p.pgrad("mask_token"), B=B, C=Dd, LIN=N, LOUT=ND,
                  LIN_S=NS, LOUT_S=NDS)
        g_lat = p.alloc("g:latent", (B, Dm, NS))
        v.linear_bwd("decoder_embed.weight", "decoder_embed.bias", latent, g_dx, Dm, Dd, NS, dx=g_lat)
        _encoder_bwd(v, s, erec, g_lat, x_dx, dx_accum=True)

    p.tape.append(backward)
    segments, bwd = finish_plan(p, layout, training, bucket_floats)
    optional = None
    if bwd is not None:
        # the two stages that bring a caller's gradient of `pred` into the feature-major layout and add it to d loss / d pred:
        # 5 HBM passes over a pred-sized tensor of zeros per step when the trainer differentiates only the loss
        i0 = next(i for i, (k, f_) in enumerate(bwd.ops) if k == "TRANSPOSE_CL" and isinstance(f_.get("X"), TRef) and f_["X"].ref == d_pred.ref)
        assert bwd.ops[i0 + 1][0] == "AXPY"
        optional = {"pred": (i0, i0 + 2)}
    return VitPlan(s, B, training, p.fwd, bwd, p.ws.mark(), p.aux.mark(), p.blob, layout, segments, outs, cur[0],
                   {"noise": noise}, B * Lp * 4, (1,), p.tensors, p.wpack.mark(), 0, x_shape,
                   douts={"loss": d_loss, "pred": d_pred}, dout_bytes=256 + B * Lp * PD * 4, want_dx=want_dx, optional_dout_ops=optional)


def plan_seg(s: SegSpec, B: int, training: bool, layout: ParamLayout | None = None, bucket_floats: int = 8 << 20,
             want_bwd: bool | None = None, want_dx: bool = False, bf16: bool = False) -> VitPlan:
    """PrithviSegmentationNet.forward (prithvi_segmentation.py:156-162).  bf16: the bf16-mixed plan (see plan_mae)."""
    with _row_align(8 if bf16 else 4):
        return _plan_seg(s, B, training, layout, bucket_floats, want_bwd, want_dx, bf16)


def _plan_seg(s: SegSpec, B: int, training: bool, layout, bucket_floats: int, want_bwd, want_dx: bool, bf16: bool) -> VitPlan:
    m = s.mae
    assert not m.decoder
    if m.img_size // m.patch_size * 16 != m.img_size:
        raise ValueError("the neck upsamples the patch grid x16: img_size must be 16 * (img_size // patch_size)")
    layout = layout or seg_layout(s)
    p = _P(s, layout, B, m.img_size, m.img_size, training, want_bwd)
    p.bf16 = bool(bf16)
    frozen = s.frozen_backbone
    v = _V(p, lambda name: not (frozen and name.startswith("backbone.")))
    Lp, Dm, E = m.num_patches, m.embed_dim, s.embed
    g = m.img_size // m.patch_size
    if m.num_frames != 1 or Lp != g * g:
        raise ValueError("PrithviSegmentationNet lays out L tokens as a (patch_height x patch_width) map: num_frames must be 1")
    x_shape = (B, m.in_chans, m.num_frames, m.img_size, m.img_size)
    x_img = TRef(D.BASE["X"], 0, x_shape, "f32", "imgs")
    noise = TRef(D.BASE["NOISE"], 0, (B, Lp), "f32", "noise")
    drop_off = (B * Lp * 4 + 255) // 256 * 256
    drop_u = TRef(D.BASE["NOISE"], drop_off, (B, s.fcn_out_channels), "f32", "drop_u")
    outs: dict = {}
    cur = [0]
    _out(outs, cur, "logits", (B, s.num_classes, m.img_size, m.img_size))
    _out(outs, cur, "mask", (B, Lp))
    _out(outs, cur, "ids_restore", (B, Lp), "i64")
    latent, N, erec = _encoder(v, m, "backbone.", x_img, noise, Lp, outs, not frozen)   # mask_ratio 0: a pure shuffle
    x_dx = TRef(D.BASE["DX"], 0, x_shape, "f32", "dx") if want_dx else None
    need_enc_bwd = p.want_bwd and (not frozen or want_dx)     # (a frozen backbone still passes the gradient through to the input)
    # neck (:66-72): drop cls, tokens -> [B][E][g][g]
    drop_idx = p.const_table([[j + 1 for j in range(Lp)] for _ in range(B)], Lp)
    t0 = p.alloc("neck_in", (B, Dm, Lp))
    NS = erec["NS"]
    p.fwd.add("TOKEN_GATHER", IN=latent, IDX=drop_idx, FILL=None, POS=None, OUT=t0, B=B, C=Dm, LIN=N, LOUT=Lp, POS_BY_SRC=0, POS_OFF=0,
              LIN_S=NS, LOUT_S=Lp)
    a0 = Act(t0, Dm, g, g, needs_grad=need_enc_bwd)
    nk = "neck.feature_pyramid_net."

    def norm2d_gelu(prefix: str, src: Act) -> Act:
        """Norm2d (LayerNorm over channels, eps 1e-6) -> GELU.  The GELU is materialised once (one extra write + read of the
        map): as a load prologue of the following ConvTranspose its erf polynomial was re-evaluated for each of the 24
        output-channel tiles of the forward conv and of the weight gradient (54 instead of 108 TF/s on the 112x112 stage)."""
        C, HW = src.C, src.H * src.W
        y, mr = v.ln_fwd(prefix, src.raw, C, HW, 1e-6)
        ga = p.alloc("gelu:" + prefix, (B, C, src.H, src.W))
        p.fwd.add("ACT_FWD", X=y, Y=ga, COUNT=B * C * HW, ACT=D.ACT_GELU)
        out = Act(ga, C, src.H, src.W)

        def backward():
            gq = out.grad                                           # w.r.t. gelu(y)
            p.bwd.add("ACT_BWD", G=gq, X=y, COUNT=B * C * HW, ACT=D.ACT_GELU)
            gx = p.grad_of(src, prefix + ".src")
            v.ln_bwd(prefix, gq, src.raw, mr, gx, C, HW, accum=int(src.grad_init))
            src.grad_init = True

        p.tape.append(backward)
        return out

    def encoder_tail_backward():
        if not need_enc_bwd:
            return
        g_lat = p.alloc("g:latent", (B, Dm, NS))
        p.bwd.add("TOKEN_SCATTER", DOUT=a0.grad, IDX=drop_idx, DIN=g_lat, DFILL=None, B=B, C=Dm, LIN=N, LOUT=Lp, LIN_S=NS, LOUT_S=Lp)
        _encoder_bwd(v, m, erec, g_lat, x_dx)

    p.tape.append(encoder_tail_backward)
    a = conv_transpose(p, nk + "0.weight", nk + "0.bias", a0, E)
    a = norm2d_gelu(nk + "1.ln", a)
    a = conv_transpose(p, nk + "3.weight", nk + "3.bias", a, E)
    a = conv_transpose(p, nk + "4.weight", nk + "4.bias", a, E)
    a = norm2d_gelu(nk + "5.ln", a)
    a = conv_transpose(p, nk + "7.weight", nk + "7.bias", a, E)
    # head (:90-111)
    idx = 0
    for _ in range(s.fcn_num_convs):
        a = conv_bn(p, f"head.net.{idx}.weight", f"head.net.{idx + 1}", [a], s.fcn_out_channels, 3, 1, False, D.PRO_RELU, 1e-5, 0.1,
                    bias=f"head.net.{idx}.bias")
        idx += 3
    idx += 1
    Cf, Hh = a.C, m.img_size
    gate = None
    if training and s.fcn_dropout > 0:
        gate = p.alloc("drop_gate", (B, Cf))
        p.fwd.add("DROP_GATE", U=drop_u, GATE=gate, COUNT=B * Cf, P=s.fcn_dropout)
        a.gate = gate
        a.mulbc = gate         # the BatchNorm backward of the producer multiplies the incoming gradient by the gate
    wname, bname = f"head.net.{idx}.weight", f"head.net.{idx}.bias"
    wp, MP = p.pack_weight("fwd", wname, s.num_classes, Cf, 1, Cf, 1, 1, 0)
    p.fwd.add("CONV", X1=a.raw, BNV1=a.bnv, GATE1=gate, X2=None, BNV2=None, WT=wp, BIAS=p.param(bname), Y=outs["logits"], STATS=None,
              RES=None, B=B, C1=Cf, C2=0, H=Hh, W=Hh, M=s.num_classes, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=Hh, WO=Hh,
              PRO1=a.pro, PRO2=0, MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=s.num_classes, NREP=1)
    head_in = a

    def out_backward():
        dY = TRef(D.BASE["DOUT"], 0, (B, s.num_classes, Hh, Hh), "f32", "dlogits")
        _conv_dgrad_wgrad(p, wname, dY, [head_in], s.num_classes, 1, 1, 0, 0, Hh, Hh, bname)

    p.tape.append(out_backward)
    segments, bwd = finish_plan(p, layout, training, bucket_floats)
    lo = 0
    if frozen:
        lo = min(off for name, (off, _) in layout.params.items() if not name.startswith("backbone."))
    noise_bytes = drop_off + B * s.fcn_out_channels * 4
    return VitPlan(s, B, training, p.fwd, bwd, p.ws.mark(), p.aux.mark(), p.blob, layout, segments, outs, cur[0],
                   {"noise": noise, "drop_u": drop_u}, noise_bytes, (B, s.num_classes, Hh, Hh), p.tensors, p.wpack.mark(), lo, x_shape,
                   want_dx=want_dx)


# ------------------------------------------------------------------------------------------------------------------
# Separately callable methods of MaskedAutoencoderViT (prithvi.py:258-350): each is its own small plan over the SAME flat
# parameter buffer and the same stage kernels as the fused forward; tensors cross the API boundary token-major, as in the
# reference, and are re-laid out feature-major by TRANSPOSE_CL stages at either end.
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class MethodPlan:
    """A VitPlan with several inputs / differentiable outputs (engine: vit_engine.run_method)."""
    spec: object
    B: int
    training: bool
    fwd: Program
    bwd: Program | None
    ws_bytes: int
    aux_bytes: int
    const_table: list
    layout: ParamLayout
    bwd_param_marks: list
    inputs: dict             # name -> TRef in the X base (packed input buffer)
    x_bytes: int
    outputs: dict            # name -> TRef in the OUT base
    out_bytes: int
    noise: dict
    noise_bytes: int
    douts: dict              # differentiable output name -> TRef in the DOUT base (upstream gradients, packed)
    dout_bytes: int
    dins: dict               # differentiable input name -> TRef in the DX base (gradients w.r.t. inputs)
    dx_bytes: int
    tensors: dict
    wpack_bytes: int = 0
    trainable_lo: int = 0


class _Packer:
    def __init__(self, base: str):
        self.base, self.off, self.refs = base, 0, {}

    def add(self, name: str, shape: tuple, dtype: str = "f32") -> TRef:
        isz = {"f32": 4, "i64": 8, "i32": 4}[dtype]
        t = TRef(D.BASE[self.base], self.off, tuple(shape), dtype, name)
        self.refs[name] = t
        self.off += (_numel(shape) * isz + 255) // 256 * 256
        return t


def _method_plan(p: _P, s, B, want_bwd, layout, xin: _Packer, outs: _Packer, noise: _Packer, douts: _Packer, dins: _Packer,
                 bucket_floats: int) -> MethodPlan:
    segments, bwd = finish_plan(p, layout, want_bwd, bucket_floats)
    return MethodPlan(s, B, want_bwd, p.fwd, bwd, p.ws.mark(), p.aux.mark(), p.blob, layout, segments, xin.refs, max(xin.off, 256),
                      outs.refs, max(outs.off, 256), noise.refs, max(noise.off, 4), douts.refs, max(douts.off, 256), dins.refs,
                      max(dins.off, 256), p.tensors, p.wpack.mark(), 0)


def plan_mae_encoder(s: MaeSpec, B: int, mask_ratio: float, want_bwd: bool, layout: ParamLayout, want_dx: bool = False,
                     bucket_floats: int = 8 << 20) -> MethodPlan:
    """forward_encoder(x, mask_ratio) -> (latent [B, 1 + keep, D], mask, ids_restore)   (prithvi.py:285-305)."""
    p = _P(s, layout, B, s.img_size, s.img_size, want_bwd)
    v = _V(p, lambda name: True)
    Lp, Dm = s.num_patches, s.embed_dim
    keep = int(Lp * (1 - mask_ratio))
    xin, outs, nz, douts, dins = _Packer("X"), _Packer("OUT"), _Packer("NOISE"), _Packer("DOUT"), _Packer("DX")
    x_img = xin.add("x", (B, s.in_chans, s.num_frames, s.img_size, s.img_size))
    noise = nz.add("noise", (B, Lp))
    N = 1 + keep
    out_lat = outs.add("latent", (B, N, Dm))
    o = {"mask": outs.add("mask", (B, Lp)), "ids_restore": outs.add("ids_restore", (B, Lp), "i64")}
    latent, N2, erec = _encoder(v, s, "", x_img, noise, keep, o, True)
    assert N2 == N
    NS = erec["NS"]
    p.fwd.add("TRANSPOSE_CL", X=latent, Y=out_lat, B=B, C=Dm, L=NS, L_OFF=0, LOUT=N)
    d_lat = douts.add("latent", (B, N, Dm))
    d_x = dins.add("x", tuple(x_img.shape)) if want_dx else None

    def backward():
        g_lat = p.alloc("g:latent", (B, Dm, NS))
        # token-major gradient [B][N][Dm] -> feature-major rows of NS floats with zero padding columns
        p.bwd.add("TRANSPOSE_CL", X=d_lat, Y=g_lat, B=B, C=N, L=Dm, L_OFF=0, LOUT=Dm, YS=NS, Y_OFF=0)
        _encoder_bwd(v, s, erec, g_lat, d_x)

    p.tape.append(backward)
    return _method_plan(p, s, B, want_bwd, layout, xin, outs, nz, douts, dins, bucket_floats)


def plan_mae_decoder(s: MaeSpec, B: int, N: int, want_bwd: bool, layout: ParamLayout, bucket_floats: int = 8 << 20) -> MethodPlan:
    """forward_decoder(x [B, N, D], ids_restore [B, L]) -> pred [B, L, patch_dim]   (prithvi.py:307-331)."""
    assert s.decoder
    p = _P(s, layout, B, s.img_size, s.img_size, want_bwd)
    v = _V(p, lambda name: True)
    Lp, Dm, Dd, PD = s.num_patches, s.embed_dim, s.decoder_embed_dim, s.patch_dim
    keep = N - 1
    xin, outs, nz, douts, dins = _Packer("X"), _Packer("OUT"), _Packer("NOISE"), _Packer("DOUT"), _Packer("DX")
    x_tm = xin.add("x", (B, N, Dm))
    ids = xin.add("ids_restore", (B, Lp), "i64")
    out_pred = outs.add("pred", (B, Lp, PD))
    NS, ND = row_stride(N), Lp + 1
    NDS = row_stride(ND)
    latent = p.alloc("latent_fm", (B, Dm, NS))
    p.fwd.add("TRANSPOSE_CL", X=x_tm, Y=latent, B=B, C=N, L=Dm, L_OFF=0, LOUT=Dm, YS=NS, Y_OFF=0)
    dec_idx = p.alloc("dec_idx", (B, 1 + Lp), "i32")
    p.fwd.add("IDS_TO_DEC_IDX", IDS=ids, DEC_IDX=dec_idx, B=B, L=Lp, KEEP=keep)
    dx = v.linear_fwd("decoder_embed.weight", "decoder_embed.bias", latent, Dm, Dd, NS)
    y0 = p.alloc("y0", (B, Dd, NDS))
    p.fwd.add("TOKEN_GATHER", IN=dx, IDX=dec_idx, FILL=p.param("mask_token"), POS=p.param("decoder_pos_embed"), OUT=y0,
              B=B, C=Dd, LIN=N, LOUT=ND, POS_BY_SRC=0, POS_OFF=0, LIN_S=NS, LOUT_S=NDS)
    drecs = []
    y = y0
    for i in range(s.decoder_depth):
        y, r = v.block_fwd(f"decoder_blocks.{i}", y, Dd, s.decoder_num_heads, int(Dd * s.mlp_ratio), ND)
        drecs.append(r)
    yn, mrd = v.ln_fwd("decoder_norm", y, Dd, NDS, 1e-5)
    pred_fm = v.linear_fwd("decoder_pred.weight", "decoder_pred.bias", yn, Dd, PD, NDS)
    p.fwd.add("TRANSPOSE_CL", X=pred_fm, Y=out_pred, B=B, C=PD, L=NDS, L_OFF=1, LOUT=Lp)
    d_pred = douts.add("pred", (B, Lp, PD))
    d_x = dins.add("x", (B, N, Dm))

    def backward():
        g_pred = p.alloc("g:pred", (B, PD, NDS))
        # [B][Lp][PD] -> [B][PD][NDS]: tokens land in columns 1 .. Lp, the cls column and the padding stay zero
        p.bwd.add("TRANSPOSE_CL", X=d_pred, Y=g_pred, B=B, C=Lp, L=PD, L_OFF=0, LOUT=PD, YS=NDS, Y_OFF=1)
        g_yn = p.alloc("g:yn", (B, Dd, NDS))
        v.linear_bwd("decoder_pred.weight", "decoder_pred.bias", yn, g_pred, Dd, PD, NDS, dx=g_yn)
        g = p.alloc("g:dec", (B, Dd, NDS))
        v.ln_bwd("decoder_norm", g_yn, y, mrd, g, Dd, NDS, accum=0, dsum=v.top_fc2_bias(drecs))
        g = v.blocks_bwd(drecs, g)
        g_dx = p.alloc("g:dx", (B, Dd, NS))
        p.bwd.add("TOKEN_SCATTER", DOUT=g, IDX=dec_idx, DIN=g_dx, DFILL=p.pgrad("mask_token"), B=B, C=Dd, LIN=N, LOUT=ND,
                  LIN_S=NS, LOUT_S=NDS)
        g_lat = p.alloc("g:latent", (B, Dm, NS))
        v.linear_bwd("decoder_embed.weight", "decoder_embed.bias", latent, g_dx, Dm, Dd, NS, dx=g_lat)
        p.bwd.add("TRANSPOSE_CL", X=g_lat, Y=d_x, B=B, C=Dm, L=NS, L_OFF=0, LOUT=N)

    p.tape.append(backward)
    return _method_plan(p, s, B, want_bwd, layout, xin, outs, nz, douts, dins, bucket_floats)


def plan_mae_loss(s: MaeSpec, B: int, want_bwd: bool, layout: ParamLayout) -> MethodPlan:
    """forward_loss(imgs, pred [B, L, patch_dim], mask [B, L]) -> loss   (prithvi.py:333-350)."""
    p = _P(s, layout, B, s.img_size, s.img_size, want_bwd)
    Lp, PD = s.num_patches, s.patch_dim
    LS = row_stride(Lp)
    xin, outs, nz, douts, dins = _Packer("X"), _Packer("OUT"), _Packer("NOISE"), _Packer("DOUT"), _Packer("DX")
    imgs = xin.add("imgs", (B, s.in_chans, s.num_frames, s.img_size, s.img_size))
    pred = xin.add("pred", (B, Lp, PD))
    mask = xin.add("mask", (B, Lp))
    loss = outs.add("loss", (1,))
    pred_fm = p.alloc("pred_fm", (B, PD, LS))
    p.fwd.add("TRANSPOSE_CL", X=pred, Y=pred_fm, B=B, C=Lp, L=PD, L_OFF=0, LOUT=PD, YS=LS, Y_OFF=0)
    acc = p.aux.alloc("mae_acc", (2,), "f64")
    geo = dict(B=B, C=s.in_chans, T=s.num_frames, H=s.img_size, W=s.img_size, P=s.patch_size, TUB=s.tubelet_size, LP=LS, L_OFF=0,
               NORM_PIX=int(s.norm_pix_loss))
    p.fwd.add("MAE_LOSS_FWD", PRED=pred_fm, IMGS=imgs, MASK=mask, LOSS=loss, ACC=acc, **geo)
    gout = douts.add("loss", (1,))
    d_pred = dins.add("pred", (B, Lp, PD))

    def backward():
        g = p.alloc("g:pred", (B, PD, LS))
        p.bwd.add("MAE_LOSS_BWD", PRED=pred_fm, IMGS=imgs, MASK=mask, ACC=acc, GOUT=gout, DPRED=g, **geo)
        p.bwd.add("TRANSPOSE_CL", X=g, Y=d_pred, B=B, C=PD, L=LS, L_OFF=0, LOUT=Lp)

    p.tape.append(backward)
    return _method_plan(p, s, B, want_bwd, layout, xin, outs, nz, douts, dins, 8 << 20)


def plan_random_masking(s: MaeSpec, B: int, L: int, Dm: int, mask_ratio: float, want_bwd: bool, layout: ParamLayout) -> MethodPlan:
    """random_masking(x [B, L, D], mask_ratio) -> (x_masked [B, keep, D], mask, ids_restore)   (prithvi.py:258-283)."""
    p = _P(s, layout, B, s.img_size, s.img_size, want_bwd)
    keep = int(L * (1 - mask_ratio))
    if keep < 1:
        raise ValueError("random_masking: nothing would be kept")
    xin, outs, nz, douts, dins = _Packer("X"), _Packer("OUT"), _Packer("NOISE"), _Packer("DOUT"), _Packer("DX")
    x_tm = xin.add("x", (B, L, Dm))
    noise = nz.add("noise", (B, L))
    out_xm = outs.add("x_masked", (B, keep, Dm))
    out_mask = outs.add("mask", (B, L))
    out_ids = outs.add("ids_restore", (B, L), "i64")
    x_fm = p.alloc("x_fm", (B, Dm, L))
    p.fwd.add("TRANSPOSE_CL", X=x_tm, Y=x_fm, B=B, C=L, L=Dm, L_OFF=0, LOUT=Dm)
    enc_idx = p.alloc("enc_idx", (B, 1 + keep), "i32")
    dec_idx = p.alloc("dec_idx", (B, 1 + L), "i32")
    p.fwd.add("MAE_MASK_INDEX", NOISE=noise, IDS_RESTORE=out_ids, MASK=out_mask, ENC_IDX=enc_idx, DEC_IDX=dec_idx, B=B, L=L, KEEP=keep)
    xm_fm = p.alloc("xm_fm", (B, Dm, 1 + keep))      # column 0 = the (absent) cls slot of the index table, dropped below
    p.fwd.add("TOKEN_GATHER", IN=x_fm, IDX=enc_idx, FILL=None, POS=None, OUT=xm_fm, B=B, C=Dm, LIN=L, LOUT=1 + keep, POS_BY_SRC=0,
              POS_OFF=0, LIN_S=L, LOUT_S=1 + keep)
    p.fwd.add("TRANSPOSE_CL", X=xm_fm, Y=out_xm, B=B, C=Dm, L=1 + keep, L_OFF=1, LOUT=keep)
    d_xm = douts.add("x_masked", (B, keep, Dm))
    d_x = dins.add("x", (B, L, Dm))

    def backward():
        g_fm = p.alloc("g:xm", (B, Dm, 1 + keep))
        p.bwd.add("TRANSPOSE_CL", X=d_xm, Y=g_fm, B=B, C=keep, L=Dm, L_OFF=0, LOUT=Dm, YS=1 + keep, Y_OFF=1)
        gx_fm = p.alloc("g:x", (B, Dm, L))
        p.bwd.add("TOKEN_SCATTER", DOUT=g_fm, IDX=enc_idx, DIN=gx_fm, DFILL=None, B=B, C=Dm, LIN=L, LOUT=1 + keep, LIN_S=L, LOUT_S=1 + keep)
        p.bwd.add("TRANSPOSE_CL", X=gx_fm, Y=d_x, B=B, C=Dm, L=L, L_OFF=0, LOUT=L)

    p.tape.append(backward)
    return _method_plan(p, s, B, want_bwd, layout, xin, outs, nz, douts, dins, 8 << 20)
