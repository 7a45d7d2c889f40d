"""Gradients that leave the network through its INPUT and through auxiliary outputs, as torch autograd gives them for the
reference's modules (the boundary is `net(x)` + `loss.backward()`):
  * x.requires_grad -> x.grad for EfficientnetUnet (train and eval mode), MaskedAutoencoderViT and PrithviSegmentationNet
    (frozen backbone included: the gradient passes through frozen weights), against float64 oracle autograd;
  * the gradient through `pred` of MaskedAutoencoderViT.forward (the trainer only uses the loss; torch differentiates
    whatever the caller builds on pred)."""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import detgen, losses_ref
from oracle import efficientnet_unet_ref as R
from oracle import prithvi_ref as P
from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
from tests.helpers import MAE_CASES, mae_inputs, rel_err, seg_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("train", [True, False])
def test_unet_input_gradient(train):
    from s2lc_amd.losses import FocalLoss
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet

    version, C, H, B, ncls = "b0", 5, 64, 2, 4
    net = R.build(version, C, ncls, drop_connect_rate=0.25)
    sd = detgen.fill_state(R.state_shapes(net), seed=51)
    model = EfficientnetUnet(EfficientNetConfig(version, C, ncls, class_distribution=[0.25] * 4, drop_connect_rate=0.25))
    model.load_state_dict(sd)
    model.to(DEV).train(train)
    x = detgen.normal("dx.x", (B, C, H, H), seed=51)
    y = detgen.labels("dx.y", (B, H, H), ncls, seed=51)
    noise = detgen.uniform("dx.dc", (len(net.blocks), B), 0.0, 1.0, seed=51)
    model.drop_connect_noise = noise if train else None
    xg = x.to(DEV).requires_grad_(True)
    loss = FocalLoss(torch.ones(ncls), 2.0, 0.0, ignore_index=0)(model(xg), y.to(DEV))
    loss.backward()
    assert xg.grad is not None and model.out_conv1x1.weight.grad is not None
    sd64 = {k: (v.detach().double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    ref = R.unet_forward(sd64, net, x64, training=train, dc_noise=noise.double() if train else None, new_buffers={})
    losses_ref.focal(ref, y, torch.ones(ncls, dtype=torch.float64), 2.0, 0.0, ignore_index=0).backward()
    e = rel_err(xg.grad.cpu().numpy(), x64.grad.numpy())
    # train-mode BatchNorm on tiny maps amplifies fp32 noise (tests/test_plan_cpu.py); eval mode is well conditioned
    assert e < (3e-2 if train else 1e-3), e
    # a frozen model still differentiates w.r.t. its input
    for p in model.parameters():
        p.requires_grad_(False)
        p.grad = None
    xg2 = x.to(DEV).requires_grad_(True)
    FocalLoss(torch.ones(ncls), 2.0, 0.0, ignore_index=0)(model(xg2), y.to(DEV)).backward()
    assert rel_err(xg2.grad.cpu().numpy(), xg.grad.cpu().numpy()) < 1e-5
    assert model.out_conv1x1.weight.grad is None


@pytest.mark.parametrize("norm_pix", [False, True])
def test_mae_input_gradient_and_gradient_through_pred(norm_pix):
    """norm_pix_loss: the target's per-patch standardisation depends on the images too (prithvi.py:341-344)"""
    tag = "small_t3_bs2"
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    cfg.norm_pix_loss = norm_pix
    model = MaskedAutoencoderViT(**{**MAE_CASES[tag][0], "norm_pix_loss": norm_pix})
    model.load_state_dict(sd)
    model.to(DEV)
    model.masking_noise = noise
    xg = x.to(DEV).requires_grad_(True)
    loss, pred, mask = model(xg, mask_ratio=ratio)
    w = torch.randn(pred.shape, generator=torch.Generator().manual_seed(7)).to(DEV)
    (loss + 0.01 * (pred * w).sum()).backward()       # a caller that builds on pred as well as on the loss
    sd64 = {k: v.detach().double().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    l64, p64, _ = P.mae_forward(sd64, cfg, x64, ratio, noise.double())
    (l64 + 0.01 * (p64 * w.cpu().double()).sum()).backward()
    assert rel_err(xg.grad.cpu().numpy(), x64.grad.numpy()) < 2e-3
    named = dict(model.named_parameters())
    scale = max(v.grad.abs().max().item() for v in sd64.values() if v.grad is not None)
    for name in ("decoder_pred.weight", "decoder_blocks.0.mlp.fc1.weight", "blocks.1.attn.qkv.weight", "patch_embed.proj.weight", "mask_token"):
        ref = sd64[name].grad
        err = (named[name].grad.cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3 * scale)
        assert err < 2e-3, (name, err)
    # forward_encoder alone: gradient of the latent w.r.t. the images
    xg2 = x.to(DEV).requires_grad_(True)
    latent, _, _ = model.forward_encoder(xg2, ratio)
    wl = torch.randn(latent.shape, generator=torch.Generator().manual_seed(8)).to(DEV)
    (latent * wl).sum().backward()
    x64b = x.double().requires_grad_(True)
    lat64, _, _ = P.forward_encoder({k: v.detach() for k, v in sd64.items()}, cfg, x64b, ratio, noise.double())
    (lat64 * wl.cpu().double()).sum().backward()
    assert rel_err(xg2.grad.cpu().numpy(), x64b.grad.numpy()) < 2e-3


@pytest.mark.parametrize("tag", ["small_train_frozen", "small_train_unfrozen"])
def test_seg_input_gradient(tag):
    from s2lc_amd.losses import CrossEntropyLoss
    from tests.test_prithvi_gpu import _seg_model

    net, cfg, sd, x, y, noise, drop_u, train = _seg_model(tag)
    net.to(DEV).train(train)
    net.masking_noise, net.dropout_noise = noise, drop_u
    xg = x.to(DEV).requires_grad_(True)
    CrossEntropyLoss(ignore_index=0)(net(xg), y.to(DEV)).backward()
    sd64 = {k: (v.detach().double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    logits64 = P.seg_forward(sd64, cfg, x64, noise.double(), training=train, drop_u=drop_u.double(), new_buffers={})
    losses_ref.cross_entropy(logits64, y, ignore_index=0).backward()
    assert xg.grad is not None
    assert rel_err(xg.grad.cpu().numpy(), x64.grad.numpy()) < 2e-2      # (train-mode BatchNorm in the head)
    if cfg.frozen_backbone:
        assert net.backbone.blocks[0].attn.qkv.weight.grad is None
