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


@pytest.mark.parametrize("tag,wide,norm_pix", [("small_bs2", True, False), ("small_t3_bs2", True, True), ("small_bs2", False, False)])
def test_mae_programs_match_oracle(tag, wide, norm_pix):
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    cfg.norm_pix_loss = norm_pix
    B = x.shape[0]
    spec = _mae_spec(cfg)
    plan = V.plan_mae(spec, B, ratio, True)
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
