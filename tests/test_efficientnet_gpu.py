"""EfficientNet as a separately callable module (reference src/modules/efficientnet_unet.py:179-263):
`encode(x) -> (x, feature_maps)` and `forward(x) -> logits`, standalone and as `EfficientnetUnet.encoder`, forward values and
the gradients torch autograd gives (parameters, input), against the float64 oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import s2lc_amd  # noqa: F401
from oracle import detgen
from oracle import efficientnet_unet_ref as R
from s2lc_amd.modules.efficientnet_unet import EfficientNet, EfficientNetConfig, EfficientnetUnet
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, C, H, NCLS, P_DROP = 2, 5, 96, 3, 0.25


def _state(seed):
    net = R.build("b0", C, NCLS, drop_connect_rate=0.25)
    sd = detgen.fill_state(R.state_shapes(net), seed=seed)
    sd["encoder.fc.3.weight"] = detgen.normal("fc.w", (NCLS, 1280), seed=seed) * 0.05
    sd["encoder.fc.3.bias"] = detgen.normal("fc.b", (NCLS,), seed=seed) * 0.05
    return net, sd


def _oracle(net, sd, x, train, dc, du, classifier):
    sdd = {k: (v.detach().double().requires_grad_(not k.endswith(("running_mean", "running_var"))) if v.dtype.is_floating_point else v)
           for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    newbuf = {}
    hx, fmaps = R.encode(sdd, R._BN(sdd, train, newbuf), net, x64, dc.double() if train else None)
    if not classifier:
        return sdd, x64, fmaps, newbuf
    pooled = hx.mean(dim=(2, 3))
    if train:
        pooled = pooled * (du.double() >= P_DROP) / (1.0 - P_DROP)
    return sdd, x64, [F.linear(pooled, sdd["encoder.fc.3.weight"], sdd["encoder.fc.3.bias"])], newbuf


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("nested", [False, True])
def test_encode_and_forward_match_oracle(train, nested):
    net, sd = _state(61)
    cfg = EfficientNetConfig("b0", C, NCLS, class_distribution=[1.0 / NCLS] * NCLS, drop_connect_rate=0.25, dropout_rate=P_DROP)
    if nested:
        owner = EfficientnetUnet(cfg)
        owner.load_state_dict({k: v for k, v in sd.items() if not k.startswith("encoder.fc")}, strict=False)
        owner.encoder.fc[3].weight.data.copy_(sd["encoder.fc.3.weight"])
        owner.encoder.fc[3].bias.data.copy_(sd["encoder.fc.3.bias"])
        owner.to(DEV).train(train)
        enc = owner.encoder
    else:
        owner = enc = EfficientNet(cfg)
        enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
        enc.to(DEV).train(train)
    x = detgen.normal("en.x", (B, C, H, H), seed=61)
    dc = detgen.uniform("en.dc", (len(net.blocks), B), 0.0, 1.0, seed=61)
    du = detgen.uniform("en.du", (B, 1280), 0.0, 1.0, seed=62)
    owner.drop_connect_noise = dc if train else None
    enc.dropout_noise = du if train else None
    tol_v, tol_g = (2e-3, 5e-2) if train else (1e-4, 1e-3)     # train-mode BN on 3x3 maps of 2 samples: see tests/test_plan_cpu.py
    pre = "encoder." if nested else ""
    for classifier in (False, True):
        for p in owner.parameters():
            p.grad = None
        nbt0 = enc.stem[1].num_batches_tracked.item()
        xg = x.to(DEV).requires_grad_(True)
        if classifier:
            outs = [enc(xg)]
        else:
            hx, outs = enc.encode(xg)
            assert hx is outs[0] or torch.equal(hx, outs[0])
            assert len(outs) == 5
        sdd, x64, refs, newbuf = _oracle(net, sd, x, train, dc, du, classifier)
        tot, tot64 = 0, 0
        for j, (o, r) in enumerate(zip(outs, refs)):
            assert tuple(o.shape) == tuple(r.shape)
            assert rel_err(o.detach().cpu().numpy(), r.detach().numpy()) < tol_v, (classifier, j)
            w = torch.randn(r.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(20 + j))
            tot = tot + (o * w.float().to(DEV)).sum()
            tot64 = tot64 + (r * w).sum()
        tot.backward()
        tot64.backward()
        assert rel_err(xg.grad.cpu().numpy(), x64.grad.numpy()) < tol_g
        named = dict(owner.named_parameters())
        worst = 0.0
        for name, ref in sdd.items():
            if not name.startswith("encoder.") or not getattr(ref, "requires_grad", False):
                continue
            mine = named[name if nested else name[len("encoder."):]].grad
            if ref.grad is None:
                assert mine is None or mine.abs().max().item() == 0, name
                continue
            assert mine is not None, name
            worst = max(worst, (mine.cpu().double() - ref.grad).abs().max().item() / max(ref.grad.abs().max().item(), 1e-3))
        assert worst < (0.3 if train else 5e-3), worst
        if nested:       # the decoder of the owner is not part of this method
            assert owner.out_conv1x1.weight.grad is None or owner.out_conv1x1.weight.grad.abs().max().item() == 0
        if train:
            assert enc.stem[1].num_batches_tracked.item() == nbt0 + 1
            got = enc.conv_head[1].running_var.cpu().numpy()
            # the oracle's new buffers start from the ORIGINAL state: compare on the first call only
            if not classifier:
                assert rel_err(got, newbuf["encoder.conv_head.1.running_var"].numpy()) < 1e-4
        # the state for the second pass must equal the oracle's: reset the running statistics
        if train:
            owner.load_state_dict({(k if nested else k[len("encoder."):]): v for k, v in sd.items() if k.startswith("encoder.")}, strict=False)


def test_unet_forward_still_works_after_encoder_methods():
    """The encoder's method engines and the U-Net's own engine share one flat parameter buffer and one gradient buffer."""
    net, sd = _state(63)
    cfg = EfficientNetConfig("b0", C, NCLS, class_distribution=[1.0 / NCLS] * NCLS, drop_connect_rate=None)
    m = EfficientnetUnet(cfg)
    m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("encoder.fc")}, strict=False)
    m.to(DEV).eval()
    x = detgen.normal("en.x2", (B, C, H, H), seed=63).to(DEV)
    with torch.no_grad():
        y0 = m(x)
        hx, fm = m.encoder.encode(x)
        y1 = m(x)
    assert torch.equal(y0, y1)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    ref = R.unet_forward(sd64, net, x.cpu().double(), training=False)
    assert rel_err(y1.cpu().numpy(), ref.numpy()) < 1e-4
    assert [tuple(f.shape[-2:]) for f in fm] == [(3, 3), (6, 6), (12, 12), (24, 24), (48, 48)]
