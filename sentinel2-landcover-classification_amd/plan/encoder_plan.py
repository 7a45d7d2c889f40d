"""Plans for the separately callable EfficientNet (reference efficientnet_unet.py:179-263):
  encode(x)  -> (x, feature_maps): conv_head output + the first block output at each new spatial size (deepest first);
  forward(x) -> logits [B, num_classes]: encode, global average pool, Dropout, Linear (`fc`).
Same stage kernels and the same flat parameter buffer as the fused U-Net program (the encoder of an EfficientnetUnet is
addressed through the `encoder.` prefix of its owner's layout; a standalone EfficientNet owns a layout without it)."""
from __future__ import annotations

from . import opdefs as D
from .program import TRef
from .unet_plan import fold_bn_finalize, Act, ParamLayout, UnetSpec, _P, _bn_backward, _conv_dgrad_wgrad, _stats, emit_encoder, fmap_block_indices, same_pads
from .vit_plan import MethodPlan, _method_plan, _Packer


def plan_encoder(spec: UnetSpec, B: int, H: int, W: int, training: bool, layout: ParamLayout, prefix: str, classifier: bool,
                 want_bwd: bool, want_dx: bool = False, dropout_p: float = 0.0) -> MethodPlan:
    if H % 32 or W % 32:
        raise ValueError(f"EfficientNet needs H, W multiples of 32 here, got {H}x{W}")
    p = _P(spec, layout, B, H, W, training, want_bwd)
    xin, outs, nz, douts, dins = _Packer("X"), _Packer("OUT"), _Packer("NOISE"), _Packer("DOUT"), _Packer("DX")
    x_ref = xin.add("x", (B, spec.in_channels, H, W))
    n_blocks = len(spec.blocks)
    dc = nz.add("drop_connect", (n_blocks, B))         # per-block, per-sample uniforms (reference :390-398), NOISE offset 0
    assert dc.off == 0
    d_x = dins.add("x", (B, spec.in_channels, H, W)) if want_dx else None
    x_in = Act(x_ref, spec.in_channels, H, W, needs_grad=bool(want_dx), grad=d_x)
    fidx = fmap_block_indices(spec, H, W)
    fmap_refs = {}
    if not classifier:
        # output order of the reference: x first, then the feature maps deepest first
        hh, ww = same_pads(H, 3, 2)[0], same_pads(W, 3, 2)[0]
        sizes = []
        for b in spec.blocks:
            hh, ww = same_pads(hh, b.kernel, b.stride)[0], same_pads(ww, b.kernel, b.stride)[0]
            sizes.append((hh, ww))
        out_x = outs.add("x", (B, spec.head_out, sizes[-1][0], sizes[-1][1]))
        for k, i in enumerate(fidx):
            fmap_refs[i] = outs.add(f"f{k}", (B, spec.blocks[i].cout, sizes[i][0], sizes[i][1]))
    head, fmaps, block_outs = emit_encoder(p, spec, x_in, prefix, fmap_refs)
    HWh = head.H * head.W
    if not classifier:
        p.fwd.add("ACT_FWD", X=head.raw, Y=out_x, BNV=head.bnv, COUNT=B * head.C * HWh, ACT=D.ACT_SILU, C=head.C, HW=HWh)
        if want_bwd:
            # upstream gradients arrive in DOUT with the layout of OUT; each block-output gradient buffer IS its DOUT region
            # (later consumers accumulate into it in place)
            head.grad = douts.add("x", tuple(out_x.shape))
            head.grad_init = True
            for k, i in enumerate(fidx):
                block_outs[i].grad = douts.add(f"f{k}", tuple(fmap_refs[i].shape))
                block_outs[i].grad_init = True
    else:
        ncls, Ch = spec.num_classes, head.C
        out_logits = outs.add("logits", (B, ncls))
        pool = p.alloc("cls_pool", (B, Ch))
        p.fwd.add("SE_POOL", Y=head.raw, BNV=head.bnv, POOL=pool, B=B, C=Ch, HW=HWh, PRO=head.pro)
        gate = None
        if training and dropout_p > 0:
            u = nz.add("dropout_u", (B, Ch))
            gate = p.alloc("cls_drop_gate", (B, Ch))
            p.fwd.add("DROP_GATE", U=u, GATE=gate, COUNT=B * Ch, P=float(dropout_p))
        wname, bname = prefix + "fc.3.weight", prefix + "fc.3.bias"
        wp, MP = p.pack_weight("fwd", wname, ncls, Ch, 1, Ch, 1, 1, 0)
        p.fwd.add("CONV", X1=pool, BNV1=None, GATE1=gate, X2=None, BNV2=None, WT=wp, BIAS=p.param(bname), Y=out_logits, STATS=None,
                  B=B, C1=Ch, C2=0, H=1, W=1, M=ncls, KH=1, KW=1, STRIDE=1, PAD_T=0, PAD_L=0, HO=1, WO=1, PRO1=D.PRO_NONE, PRO2=0,
                  MODE=D.MODE_CONV, W_SM=1, W_SK=MP, W_ST=MP, FLIP=0, BETA=0, YC=ncls, NREP=1)
        d_logits = douts.add("logits", (B, ncls)) if want_bwd else None
        pool_act = Act(pool, Ch, 1, 1, gate=gate)

        def cls_backward():
            _conv_dgrad_wgrad(p, wname, d_logits, [pool_act], ncls, 1, 1, 0, 0, 1, 1, bname)      # dW, db, d(pool * gate) -> pool_act.grad
            g_pool = pool_act.grad
            if gate is not None:
                p.bwd.add("ACT_BWD", G=g_pool, X=gate, COUNT=B * Ch, ACT=D.ACT_MUL)
            # d head activation = g_pool / HW on every pixel: the BatchNorm backward takes it as the broadcast term (the
            # pixel-wise gradient G is zero)
            zero = p.alloc("cls_zero_g", (B, Ch, HWh))
            p.bwd.add("MEMSET", DST=zero, BYTES=B * Ch * HWh * 4)
            head.grad, head.grad_init = zero, True
            head.addbc, head.addscale = g_pool, 1.0 / HWh

        p.tape.append(cls_backward)
    fold_bn_finalize(p.fwd)
    return _method_plan(p, spec, B, want_bwd, layout, xin, outs, nz, douts, dins, 8 << 20)
