"""CPU: the Prithvi planner's forward and hand-derived backward programs, run through the stage emulator in
float64 ("wide"), reproduce the oracle's outputs and its autograd gradients to ~1e-9 — this checks the planner's
algebra (residual-stream accumulation, LayerNorm / attention / GELU backward, token gather/scatter, MAE loss)
exactly, independent of fp32 noise.  An fp32 run checks the 1e-3 output bar."""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import losses_ref
from oracle import prithvi_ref as P
from s2lc_amd.plan import opdefs as D
from s2lc_amd.plan import vit_plan as V
from tests.helpers import MAE_CASES, SEG_CASES, mae_inputs, rel_err, seg_inputs
from tests.plan_harness import emulate, flat_from_state, fview, make_bases_vit, out_view


def _mae_spec(cfg: P.MaeCfg, decoder=True) -> V.MaeSpec:
    return V.MaeSpec(cfg.img_size, cfg.patch_size, cfg.num_frames, cfg.tubelet_size, cfg.in_chans, cfg.embed_dim, cfg.depth,
                     cfg.num_heads, cfg.decoder_embed_dim, cfg.decoder_depth, cfg.decoder_num_heads, cfg.mlp_ratio,
                     cfg.norm_pix_loss, decoder)


def _check_grads(plan, grads, sd64, tol, skip=()):
    scale = max(v.grad.abs().max().item() for v in sd64.values() if getattr(v, "grad", None) is not None)
    worst = 0.0
    for name, (off, shape) in plan.layout.params.items():
        g = grads[off:off + int(np.prod(shape))].view(shape)
        ref = sd64[name].grad
        if ref is None:
            assert g.abs().max() == 0, name
            continue
        if name in skip:
            assert g.abs().max() < 1e-6 * scale + 1e-12
            continue
        err = (g.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * scale)
        worst = max(worst, err)
        assert err < tol, (name, err)
    return worst


@pytest.mark.parametrize("tag,wide,norm_pix,fuse_gelu", [("small_bs2", True, False, False), ("small_t3_bs2", True, True, False), ("small_bs2", False, False, False),
                                                        ("small_bs2", True, False, True), ("small_bs2", False, False, True)])
def test_mae_programs_match_oracle(tag, wide, norm_pix, fuse_gelu, monkeypatch):
    """fuse_gelu: fc2's data gradient times gelu'(fc1 output) in the CONV epilogue (FLAG_RES_GELU_GRAD; off by default, see
    plan/vit_plan.py block_bwd) instead of a separate ACT_BWD stage - the same gradients."""
    monkeypatch.setattr(V, "FUSE_GELU_GRAD", fuse_gelu)
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    cfg.norm_pix_loss = norm_pix
    B = x.shape[0]
    spec = _mae_spec(cfg)
    plan = V.plan_mae(spec, B, ratio, True)
    n_fused = sum(1 for k, f in plan.bwd.ops if k == "CONV" and f.get("_flags", 0) & D.FLAG_RES_GELU_GRAD)
    n_act = sum(1 for k, f in plan.bwd.ops if k == "ACT_BWD")
    assert (n_fused > 0 and n_act == 0) if fuse_gelu else (n_fused == 0 and n_act > 0)
    dt = torch.float64 if wide else torch.float32
    fp, fb = flat_from_state(plan.layout, sd)
    bases = make_bases_vit(plan, fp, fb, x, noise, wide)
    emulate(plan.fwd.pack(), bases, wide)

    sd64 = {k: v.detach().double().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    loss64, pred64, mask64 = P.mae_forward(sd64, cfg, x.double(), ratio, noise.double())
    lat64, _, ids64 = P.forward_encoder(sd64, cfg, x.double(), ratio, noise.double())
    loss64.backward()
    tol = 1e-6 if wide else 2e-4   # wide: limited by the f32-rounded eps / scale constants carried in the stage records
    assert torch.equal(out_view(bases, plan.outputs["ids_restore"], wide), ids64)
    assert torch.equal(out_view(bases, plan.outputs["mask"], wide).double(), mask64)
    assert rel_err(out_view(bases, plan.outputs["pred"], wide).numpy(), pred64.detach().numpy()) < tol
    assert rel_err(out_view(bases, plan.outputs["latent"], wide).numpy(), lat64.detach().numpy()) < tol
    assert abs(out_view(bases, plan.outputs["loss"], wide).item() - loss64.item()) < tol * abs(loss64.item())
    fview(bases, "DOUT", wide)[0] = 1.0
    emulate(plan.bwd.pack(), bases, wide)
    _check_grads(plan, fview(bases, "GRADS", wide), sd64, 1e-5 if wide else 5e-3)


@pytest.mark.parametrize("tag,wide", [("small_train_unfrozen", True), ("small_train_frozen", True), ("small_eval", False),
                                      ("small_train_unfrozen", False)])
def test_seg_programs_match_oracle(tag, wide):
    cfg, sd, x, y, noise, drop_u, train = seg_inputs(tag)
    B = x.shape[0]
    spec = V.SegSpec(_mae_spec(cfg.mae, decoder=False), cfg.num_classes, cfg.fcn_out_channels, cfg.fcn_num_convs, cfg.fcn_dropout,
                     cfg.frozen_backbone)
    plan = V.plan_seg(spec, B, train)
    fp, fb = flat_from_state(plan.layout, sd)
    nz = torch.zeros(plan.noise_bytes // 4)
    nz[: noise.numel()] = noise.reshape(-1)
    doff = plan.noise["drop_u"].off // 4
    nz[doff:doff + drop_u.numel()] = drop_u.reshape(-1)
    bases = make_bases_vit(plan, fp, fb, x, nz, wide)
    emulate(plan.fwd.pack(), bases, wide)
    logits = out_view(bases, plan.outputs["logits"], wide).clone()

    sd64 = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.detach().double()
            frozen = cfg.frozen_backbone and k.startswith("backbone.")
            if train and not frozen and not k.endswith(("pos_embed", "running_mean", "running_var")):
                v.requires_grad_(True)
        sd64[k] = v
    newbuf = {}
    with torch.set_grad_enabled(train):
        logits64 = P.seg_forward(sd64, cfg, x.double(), noise.double(), training=train, drop_u=drop_u.double(), new_buffers=newbuf)
    assert rel_err(logits.numpy(), logits64.detach().numpy()) < (1e-6 if wide else 2e-4)
    if not train:
        return
    ce = losses_ref.cross_entropy(logits64, y, ignore_index=0)
    (dlogits,) = torch.autograd.grad(ce, logits64, retain_graph=True)
    ce.backward()
    fview(bases, "DOUT", wide).copy_(dlogits.reshape(-1))
    emulate(plan.bwd.pack(), bases, wide)
    # a conv bias in front of train-mode BatchNorm has an analytically zero gradient: the planner emits exact 0
    _check_grads(plan, fview(bases, "GRADS", wide), sd64, 1e-5 if wide else 2e-2, skip=("head.net.0.bias",))
    bufs = fview(bases, "BUFS", wide)
    for nm in ("running_mean", "running_var"):
        off, shape = plan.layout.bufs["head.net.1." + nm]
        assert rel_err(bufs[off:off + shape[0]].numpy(), newbuf["head.net.1." + nm].numpy()) < (1e-6 if wide else 1e-5)
    if cfg.frozen_backbone:
        assert plan.trainable_lo == plan.layout.params["neck.feature_pyramid_net.0.weight"][0]


# ---- separately callable methods (forward_encoder / forward_decoder / forward_loss / random_masking) ---------------------
def _method_bases(plan, fp, fb, inputs: dict, noise: dict, wide=True):
    """Packed X / NOISE / DOUT / DX bases of a MethodPlan for the emulator."""
    from s2lc_amd.plan import opdefs as D
    from tests.plan_harness import _bytes

    fd = torch.float64
    bases = make_bases_vit(type("P", (), dict(ws_bytes=plan.ws_bytes, aux_bytes=plan.aux_bytes, out_bytes=plan.out_bytes, noise_bytes=max(plan.noise_bytes, 4),
                                               dout_shape=(1,), const_table=plan.const_table, wpack_bytes=plan.wpack_bytes))(),
                           fp, fb, torch.zeros(1), torch.zeros(1), wide)
    k = 2
    mem = __import__("oracle.ops_ref", fromlist=["Mem"])
    for base, size in (("X", plan.x_bytes), ("DOUT", plan.dout_bytes), ("DX", plan.dx_bytes), ("NOISE", max(plan.noise_bytes, 8))):
        bases[D.BASE[base]] = torch.zeros(k * ((size + 7) // 8 * 8) + 64, dtype=torch.uint8)
    m = mem.Mem(bases, wide, (D.BASE["CONST"],))
    for name, t in inputs.items():
        r = plan.inputs[name]
        m.view(r.ref, r.shape, r.dtype).copy_(t)
    for name, t in noise.items():
        r = plan.noise[name]
        m.view(r.ref, r.shape, r.dtype).copy_(t)
    return bases, m


def test_method_plans_match_oracle_float64():
    cfg, sd, x, noise, ratio = mae_inputs("small_bs2")
    B = x.shape[0]
    spec = _mae_spec(cfg)
    layout = V.mae_layout(spec)
    fp, fb = flat_from_state(layout, sd)
    sd64 = {k: v.detach().double().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    x64, n64 = x.double(), noise.double()
    lat64, mask64, ids64 = P.forward_encoder(sd64, cfg, x64, ratio, n64)
    # --- forward_encoder: outputs + gradient of sum(latent * w) --------------------------------------------------------
    plan = V.plan_mae_encoder(spec, B, ratio, True, layout)
    bases, m = _method_bases(plan, fp, fb, {"x": x64}, {"noise": n64})
    emulate(plan.fwd.pack(), bases, True)
    o = plan.outputs
    assert torch.equal(m.view(o["ids_restore"].ref, o["ids_restore"].shape, "i64"), ids64)
    assert torch.equal(m.view(o["mask"].ref, o["mask"].shape).double(), mask64)
    assert rel_err(m.view(o["latent"].ref, o["latent"].shape).numpy(), lat64.detach().numpy()) < 1e-6
    w = torch.randn(lat64.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (lat64 * w).sum().backward()
    d = plan.douts["latent"]
    m.view(d.ref, d.shape).copy_(w)
    emulate(plan.bwd.pack(), bases, True)
    enc_names = {n for n in layout.params if not n.startswith(("decoder", "mask_token"))}
    scale = max(sd64[n].grad.abs().max().item() for n in enc_names if sd64[n].grad is not None)
    for name in enc_names:
        off, shape = layout.params[name]
        g = fview(bases, "GRADS", True)[off:off + int(np.prod(shape))].view(shape)
        ref = sd64[name].grad
        if ref is None:
            assert g.abs().max() == 0, name
        else:
            assert (g - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1e-3 * scale), name
    # --- forward_decoder: pred + gradients w.r.t. x and the decoder parameters ----------------------------------------
    for v in sd64.values():
        v.grad = None
    lat_in = lat64.detach().clone().requires_grad_(True)
    pred64 = P.forward_decoder(sd64, cfg, lat_in, ids64)
    plan = V.plan_mae_decoder(spec, B, lat_in.shape[1], True, layout)
    bases, m = _method_bases(plan, fp, fb, {"x": lat_in.detach(), "ids_restore": ids64}, {})
    emulate(plan.fwd.pack(), bases, True)
    r = plan.outputs["pred"]
    assert rel_err(m.view(r.ref, r.shape).numpy(), pred64.detach().numpy()) < 1e-6
    w = torch.randn(pred64.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    (pred64 * w).sum().backward()
    d = plan.douts["pred"]
    m.view(d.ref, d.shape).copy_(w)
    emulate(plan.bwd.pack(), bases, True)
    gx = plan.dins["x"]
    assert rel_err(m.view(gx.ref, gx.shape).numpy(), lat_in.grad.numpy()) < 1e-5
    for name in layout.params:
        if name.startswith(("decoder_blocks", "decoder_embed", "decoder_norm", "decoder_pred", "mask_token")):
            off, shape = layout.params[name]
            g = fview(bases, "GRADS", True)[off:off + int(np.prod(shape))].view(shape)
            ref = sd64[name].grad
            assert (g - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1e-9), name
    # --- forward_loss: value + gradient w.r.t. pred ---------------------------------------------------------------------
    pr = pred64.detach().clone().requires_grad_(True)
    loss64 = P.forward_loss(cfg, x64, pr, mask64)
    loss64.backward()
    plan = V.plan_mae_loss(spec, B, True, layout)
    bases, m = _method_bases(plan, fp, fb, {"imgs": x64, "pred": pr.detach(), "mask": mask64}, {})
    emulate(plan.fwd.pack(), bases, True)
    r = plan.outputs["loss"]
    assert abs(m.view(r.ref, r.shape).item() - loss64.item()) < 1e-9 * abs(loss64.item())
    d = plan.douts["loss"]
    m.view(d.ref, d.shape).fill_(1.0)
    emulate(plan.bwd.pack(), bases, True)
    gp = plan.dins["pred"]
    assert rel_err(m.view(gp.ref, gp.shape).numpy(), pr.grad.numpy()) < 1e-9
    # --- random_masking: exact indices, gathered rows, scatter backward ----------------------------------------------
    xt = torch.randn(B, cfg.num_patches, 12, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    xm64, mk64, id64 = P.random_masking(xt, 0.5, n64)
    plan = V.plan_random_masking(spec, B, cfg.num_patches, 12, 0.5, True, layout)
    bases, m = _method_bases(plan, fp, fb, {"x": xt.detach()}, {"noise": n64})
    emulate(plan.fwd.pack(), bases, True)
    o = plan.outputs
    assert torch.equal(m.view(o["x_masked"].ref, o["x_masked"].shape), xm64.detach())
    assert torch.equal(m.view(o["ids_restore"].ref, o["ids_restore"].shape, "i64"), id64)
    assert torch.equal(m.view(o["mask"].ref, o["mask"].shape).double(), mk64)
    w = torch.randn(xm64.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    (xm64 * w).sum().backward()
    d = plan.douts["x_masked"]
    m.view(d.ref, d.shape).copy_(w)
    emulate(plan.bwd.pack(), bases, True)
    gx = plan.dins["x"]
    assert torch.equal(m.view(gx.ref, gx.shape), xt.grad)


@pytest.mark.parametrize("norm_pix", [False, True])
def test_mae_input_gradient_program_float64(norm_pix):
    """want_dx: d loss / d imgs = the encoder path (inverse patchify of the patch-embed data gradient) + the loss TARGET path
    (the reference's forward_loss differentiates patchify(imgs) too); plus an upstream gradient on pred."""
    cfg, sd, x, noise, ratio = mae_inputs("small_t3_bs2")
    cfg.norm_pix_loss = norm_pix       # (the target's per-patch standardisation is differentiated too, prithvi.py:341-344)
    B = x.shape[0]
    spec = _mae_spec(cfg)
    plan = V.plan_mae(spec, B, ratio, True, want_dx=True)
    fp, fb = flat_from_state(plan.layout, sd)
    bases = make_bases_vit(plan, fp, fb, x, noise, True)
    emulate(plan.fwd.pack(), bases, True)
    sd64 = {k: v.detach().double().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    loss64, pred64, _ = P.mae_forward(sd64, cfg, x64, ratio, noise.double())
    w = torch.randn(pred64.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    (loss64 + 0.1 * (pred64 * w).sum()).backward()
    from oracle import ops_ref
    from s2lc_amd.plan import opdefs as D

    m = ops_ref.Mem(bases, True, (D.BASE["CONST"],))
    m.view(plan.douts["loss"].ref, (1,)).fill_(1.0)
    m.view(plan.douts["pred"].ref, plan.douts["pred"].shape).copy_(0.1 * w)
    emulate(plan.bwd.pack(), bases, True)
    dx = m.view((D.BASE["DX"] << 56), tuple(x.shape))
    assert (dx - x64.grad).abs().max().item() <= 1e-6 * x64.grad.abs().max().item()
    _check_grads(plan, fview(bases, "GRADS", True), sd64, 1e-5)


def test_no_main_stream_stage_overwrites_what_a_side_stream_stage_still_reads():
    """The executor orders a side-stream stage (weight / bias gradients) after earlier main-stream work only; a later main-stream
    stage that writes what it reads must carry FLAG_JOIN.  The transformer blocks accumulate the residual-stream gradient in place
    (CHAN_LN_BWD adds into the buffer the fc2 / proj weight gradients read): check the invariant on the final programs of every
    planner (U-Net: no such write exists, so no join is added either)."""
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
    from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
    from s2lc_amd.modules.prithvi_segmentation import PrithviSegmentationNet, PrithviSegmentationNetConfig
    from s2lc_amd.plan import opdefs as D
    from s2lc_amd.plan.program import TRef
    from tests.helpers import PRITHVI_SEG_SMALL, PRITHVI_SMALL

    def check(prog):
        pending, joins = [], 0
        for i, (kind, f) in enumerate(prog.ops):
            flags = f.get("_flags", 0)
            names = D.OPS[kind][0]
            writes = D.WRITES.get(kind, tuple(names))
            rng = lambda ks: [(v.base, v.off, v.off + v.nbytes) for k in ks if isinstance(v := f.get(k), TRef)]  # noqa: E731
            if flags & D.FLAG_SIDE:
                pending += rng([k for k in names if k not in writes])
                continue
            if flags & D.FLAG_JOIN:
                joins += 1
                pending = []
                continue
            for (b, lo, hi) in rng(writes):
                assert not any(b == pb and lo < phi and plo < hi for (pb, plo, phi) in pending), (i, kind, f.get("DX") or f.get("Y"))
        return joins

    mae = MaskedAutoencoderViT(**PRITHVI_SMALL)
    plan = mae._make_plan(2, True, 0.75, True)
    j = check(plan.bwd)
    # the residual-stream gradient gets a fresh buffer at every update (CHAN_LN_BWD with DXIN), so no LayerNorm backward has to
    # wait for the side stream: the only joins would be buckets' WGRAD_FINALIZE stages kept on the main stream (since round 3 they run on the
    # side stream, behind the weight gradients whose scratch they fold: none)
    assert j == sum(1 for k, f in plan.bwd.ops if k == "WGRAD_FINALIZE" and not f.get("_flags", 0) & D.FLAG_SIDE)
    lnb = [f for k, f in plan.bwd.ops if k == "CHAN_LN_BWD"]
    assert sum(1 for f in lnb if f.get("DXIN") is not None) == 2 * (PRITHVI_SMALL["depth"] + PRITHVI_SMALL["decoder_depth"])
    # the proj / fc2 bias gradients come out of those stages: one CHANNEL_SUM per block less two
    n_blocks = PRITHVI_SMALL["depth"] + PRITHVI_SMALL["decoder_depth"]
    assert sum(1 for f in lnb if f.get("DSUM") is not None) == 2 * n_blocks
    bb = MaskedAutoencoderViT(**PRITHVI_SEG_SMALL, _decoder=False, _flat=False)
    cfg = PrithviSegmentationNetConfig(num_frames=1, num_classes=4, fcn_out_channels=8, fcn_num_convs=1, fcn_dropout=0.1,
                                       frozen_backbone=False, embed_dim=32, patch_height=4, patch_width=4)
    check(PrithviSegmentationNet(cfg, backbone=bb)._make_plan(2, True, 0.0, True).bwd)
    unet = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4))
    plan = unet._make_plan(2, 64, 64, True)
    fin = [f for k, f in plan.bwd.ops if k == "WGRAD_FINALIZE"]
    assert fin and all(f.get("_flags", 0) & D.FLAG_SIDE for f in fin) and check(plan.bwd) == 0      # no join at all in the U-Net's backward
