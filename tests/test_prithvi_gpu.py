"""GPU parity of the Prithvi path through the product modules (C ABI underneath): MaskedAutoencoderViT and
PrithviSegmentationNet against the golden fixtures of the imported reference (outputs: 1e-3 relative bar of
BASELINE.json, measured ~1e-5; masks / indices bit-exact) and against float64 oracle autograd (gradients)."""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import losses_ref
from oracle import prithvi_ref as P
from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
from s2lc_amd.modules.prithvi_segmentation import PrithviSegmentationNet, PrithviSegmentationNetConfig
from tests.helpers import MAE_CASES, SEG_CASES, load, mae_inputs, rel_err, seg_inputs, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grad_check(model, sd64, tol, zero_ok=()):
    scale = max(v.grad.abs().max().item() for v in sd64.values() if getattr(v, "grad", None) is not None)
    for name, p in model.named_parameters():
        ref = sd64[name].grad
        if ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, name
            continue
        assert p.grad is not None, name
        g = p.grad.detach().cpu().double()
        if name in zero_ok:
            assert g.abs().max().item() < 1e-5 * scale
            continue
        err = (g - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * scale)
        assert err < tol, (name, err)


@pytest.mark.parametrize("tag", list(MAE_CASES))
def test_mae_matches_reference_fixture(tag):
    g = load(f"prithvi_mae_{tag}.npz")
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    args = MAE_CASES[tag][0]
    model = MaskedAutoencoderViT(**args)
    model.load_state_dict(sd)
    model.to(DEV)
    model.masking_noise = noise
    grads = MAE_CASES[tag][4]
    with torch.set_grad_enabled(grads):
        loss, pred, mask = model(x.to(DEV), mask_ratio=ratio)
    latent, mask2, ids = model.forward_encoder(x.to(DEV), ratio)
    assert np.array_equal(mask.cpu().to(torch.uint8).numpy(), g["mask"]) and torch.equal(mask, mask2)
    assert np.array_equal(ids.cpu().to(torch.int32).numpy(), g["ids_restore"])
    assert rel_err(sub(pred.cpu(), 4096), g["pred_sub"]) < 1e-3
    assert rel_err(sub(latent.cpu(), 4096), g["latent_sub"]) < 1e-3
    assert tuple(pred.shape) == (x.shape[0], cfg.num_patches, cfg.patch_dim)
    if np.isnan(g["loss"][0]):
        assert torch.isnan(loss)
    else:
        assert abs(loss.item() - g["loss"][0]) < 1e-4 * abs(g["loss"][0])
    if grads:
        loss.backward()
        sd64 = {k: v.detach().double().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
        l64, _, _ = P.mae_forward(sd64, cfg, x.double(), ratio, noise.double())
        l64.backward()
        _grad_check(model, sd64, 2e-3)
        for key in g.files:      # and the reference's own gradients
            if key.startswith("grad:"):
                got = dict(model.named_parameters())[key[5:]].grad
                assert rel_err(sub(got.cpu(), 512), g[key]) < 2e-3, key


def _seg_model(tag):
    cfg, sd, x, y, noise, drop_u, train = seg_inputs(tag)
    args = SEG_CASES[tag][0]
    bb = MaskedAutoencoderViT(**args, _decoder=False, _flat=False)
    gsz = cfg.mae.img_size // cfg.mae.patch_size
    net = PrithviSegmentationNet(PrithviSegmentationNetConfig(cfg.mae.num_frames, cfg.num_classes, cfg.fcn_out_channels, cfg.fcn_num_convs,
                                                              cfg.fcn_dropout, cfg.frozen_backbone, embed_dim=cfg.mae.embed_dim,
                                                              patch_height=gsz, patch_width=gsz), backbone=bb)
    net.load_state_dict(sd)
    net.to(DEV)
    net.masking_noise, net.dropout_noise = noise, drop_u
    return net, cfg, sd, x, y, noise, drop_u, train


@pytest.mark.parametrize("tag", list(SEG_CASES))
def test_seg_matches_reference_fixture(tag):
    g = load(f"prithvi_seg_{tag}.npz")
    net, cfg, sd, x, y, noise, drop_u, train = _seg_model(tag)
    net.train(train)
    with torch.set_grad_enabled(train):
        logits = net(x.to(DEV))
    assert rel_err(sub(logits.cpu(), 4096), g["logits_sub"]) < 1e-3
    from s2lc_amd.losses import class_mask
    got_mask = class_mask(logits).cpu().to(torch.uint8).numpy()
    ref_mask = g["mask"]
    if g["margin_min"][0] > 1e-3:
        assert np.array_equal(got_mask, ref_mask)
    else:   # identical except where the reference's own top-2 margin is inside fp32 noise
        assert (got_mask != ref_mask).mean() < 1e-4
    from s2lc_amd.losses import CrossEntropyLoss
    ce = CrossEntropyLoss(ignore_index=0)(logits, y.to(DEV))        # the PRODUCT loss (what bench.py's seg legs run), not torch's
    assert abs(ce.item() - g["loss_ce"][0]) < 2e-4 * abs(g["loss_ce"][0])
    if not train:
        return
    ce.backward()
    sd64 = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.detach().double()
            frozen = cfg.frozen_backbone and k.startswith("backbone.")
            if not frozen and not k.endswith(("pos_embed", "running_mean", "running_var")):
                v.requires_grad_(True)
        sd64[k] = v
    newbuf = {}
    lg64 = P.seg_forward(sd64, cfg, x.double(), noise.double(), training=True, drop_u=drop_u.double(), new_buffers=newbuf)
    losses_ref.cross_entropy(lg64, y, ignore_index=0).backward()
    _grad_check(net, sd64, 2e-2, zero_ok=("head.net.0.bias",))
    st = net.state_dict()
    assert rel_err(st["head.net.1.running_mean"].cpu().numpy(), g["rm:head.net.1"]) < 1e-4
    assert rel_err(st["head.net.1.running_var"].cpu().numpy(), g["rv:head.net.1"]) < 1e-4
    assert int(st["head.net.1.num_batches_tracked"]) == int(g["nbt:head.net.1"][0])
    if cfg.frozen_backbone:
        assert all(p.grad is None for n, p in net.named_parameters() if n.startswith("backbone."))


def test_mae_default_noise_and_optimizer_step():
    """Without injected noise the masking draws U[0,1) on the device; FlatAdam leaves the fixed position tables alone."""
    from s2lc_amd.optim import FlatAdam
    from tests.helpers import PRITHVI_SMALL

    torch.manual_seed(0)
    model = MaskedAutoencoderViT(**PRITHVI_SMALL).to(DEV)
    opt = FlatAdam(model, lr=1e-2, weight_decay=0.1)
    x = torch.randn(4, 3, 1, 32, 32, device=DEV)
    pos0 = model.pos_embed.detach().clone()
    w0 = model.blocks[0].attn.qkv.weight.detach().clone()
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss, pred, mask = model(x)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        assert mask.sum().item() == 4 * 12 and set(mask.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(model.pos_embed, pos0)
    assert not torch.equal(model.blocks[0].attn.qkv.weight, w0)
    assert all(np.isfinite(losses))


# ---- the separately callable methods of the reference's surface (prithvi.py:258-350) -------------------------------------
def test_random_masking_matches_reference_fixture_exactly():
    """MaskedAutoencoderViT.random_masking on the reference's injected noise: kept rows, mask and ids_restore bit-exact."""
    from oracle import detgen
    from tests.helpers import PRITHVI_SMALL

    g = load("prithvi_misc.npz")
    model = MaskedAutoencoderViT(**PRITHVI_SMALL).to(DEV)
    for tag in "abcd":
        N, L, Dm = (int(v) for v in g[f"rm:{tag}:shape"])
        r = float(g[f"rm:{tag}:ratio"][0])
        x = detgen.normal(f"rm.{tag}.x", (N, L, Dm), seed=5)
        model.masking_noise = detgen.uniform(f"rm.{tag}.n", (N, L), 0.0, 1.0, seed=5)
        xm, mask, ids = model.random_masking(x.to(DEV), r)
        assert not xm.requires_grad            # (no parameters involved: differentiable only through x)
        assert np.array_equal(xm.cpu().numpy(), g[f"rm:{tag}:xm"]), tag
        assert np.array_equal(mask.cpu().to(torch.uint8).numpy(), g[f"rm:{tag}:mask"]), tag
        assert np.array_equal(ids.cpu().to(torch.int32).numpy(), g[f"rm:{tag}:ids"]), tag
    # differentiable like the reference's gather: d(sum x_masked * w) / dx scatters w back to the kept rows
    xg = x.to(DEV).requires_grad_(True)
    xm, mask, ids = model.random_masking(xg, r)
    w = torch.randn_like(xm)
    (xm * w).sum().backward()
    xr = x.clone().requires_grad_(True)
    xm_ref, _, _ = P.random_masking(xr, r, model.masking_noise)
    (xm_ref * w.cpu()).sum().backward()
    assert torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.parametrize("tag", ["small_bs2", "small_t3_bs2"])
def test_encoder_decoder_loss_methods_compose_to_the_fused_forward(tag):
    """forward_encoder -> forward_decoder -> forward_loss (three autograd nodes, three plans) against the reference fixture
    and against the fused forward(): same loss / pred / latent, same parameter gradients."""
    g = load(f"prithvi_mae_{tag}.npz")
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    args = MAE_CASES[tag][0]
    xg = x.to(DEV)

    def fresh():
        m = MaskedAutoencoderViT(**args)
        m.load_state_dict(sd)
        m.to(DEV)
        m.masking_noise = noise
        return m

    fused = fresh()
    loss_f, pred_f, mask_f = fused(xg, mask_ratio=ratio)
    loss_f.backward()
    model = fresh()
    latent, mask, ids = model.forward_encoder(xg, ratio)
    pred = model.forward_decoder(latent, ids)
    loss = model.forward_loss(xg, pred, mask)
    assert np.array_equal(mask.cpu().to(torch.uint8).numpy(), g["mask"]) and np.array_equal(ids.cpu().to(torch.int32).numpy(), g["ids_restore"])
    assert rel_err(sub(latent.detach().cpu(), 4096), g["latent_sub"]) < 1e-3
    assert rel_err(sub(pred.detach().cpu(), 4096), g["pred_sub"]) < 1e-3
    assert abs(loss.item() - g["loss"][0]) < 1e-4 * abs(g["loss"][0])
    assert rel_err(pred.detach().cpu().numpy(), pred_f.detach().cpu().numpy()) < 1e-5
    assert abs(loss.item() - loss_f.item()) < 1e-6 * abs(loss_f.item())
    loss.backward()
    torch.cuda.synchronize()
    gf, gm = fused._grad_buffer(), model._grad_buffer()
    scale = gf.abs().max().item()
    assert (gf - gm).abs().max().item() <= 1e-4 * scale, (gf - gm).abs().max().item() / scale
    for key in g.files:      # and the reference's own gradients
        if key.startswith("grad:"):
            got = dict(model.named_parameters())[key[5:]].grad
            assert rel_err(sub(got.cpu(), 512), g[key]) < 2e-3, key
    # gradient w.r.t. pred through forward_loss alone, and w.r.t. the latent through forward_decoder alone, vs oracle autograd
    sd64 = {k: v.detach().double() for k, v in sd.items()}
    pr = pred.detach().clone().requires_grad_(True)
    model.forward_loss(xg, pr, mask).backward()
    pr64 = pred.detach().cpu().double().requires_grad_(True)
    P.forward_loss(cfg, x.double(), pr64, mask.cpu().double()).backward()
    assert rel_err(pr.grad.cpu().numpy(), pr64.grad.numpy()) < 1e-4
    lat = latent.detach().clone().requires_grad_(True)
    w = torch.randn_like(pred)
    (model.forward_decoder(lat, ids) * w).sum().backward()
    lat64 = latent.detach().cpu().double().requires_grad_(True)
    (P.forward_decoder(sd64, cfg, lat64, ids.cpu()) * w.cpu().double()).sum().backward()
    assert rel_err(lat.grad.cpu().numpy(), lat64.grad.numpy()) < 2e-3


def test_forward_encoder_on_a_decoderless_backbone():
    """load_prithvi(no_decoder=True) yields a decoder-less model; the reference calls its forward_encoder
    (prithvi_segmentation.py:159)."""
    cfg, sd, x, noise, ratio = mae_inputs("small_r0_bs2")
    g = load("prithvi_mae_small_r0_bs2.npz")
    args = MAE_CASES["small_r0_bs2"][0]
    model = MaskedAutoencoderViT(**args, _decoder=False)
    model.load_state_dict({k: v for k, v in sd.items() if k in model.state_dict()})
    model.to(DEV)
    model.masking_noise = noise
    with torch.no_grad():
        latent, mask, ids = model.forward_encoder(x.to(DEV), 0.0)
    assert np.array_equal(ids.cpu().to(torch.int32).numpy(), g["ids_restore"])
    assert rel_err(sub(latent.cpu(), 4096), g["latent_sub"]) < 1e-3
    with pytest.raises(RuntimeError, match="without its decoder"):
        model.forward_decoder(latent, ids)
