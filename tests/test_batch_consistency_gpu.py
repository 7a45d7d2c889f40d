"""GPU, BASELINE-sized shapes: size-independent property — in eval mode a sample's logits do not depend on what else is in
the batch.  Catches batch-index / large-tensor addressing errors the small parity cases cannot see (e.g. 32-bit buffer
offsets on the 2.3 GiB activations of the Prithvi head at batch 15)."""
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import detgen
from oracle import efficientnet_unet_ref as R
from tests.helpers import PRITHVI_FULL, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_unet_b5_256x13_eval_batch_independent():
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet

    net = R.build("b5", 13, 4)
    sd = detgen.fill_state(R.state_shapes(net), seed=41)
    model = EfficientnetUnet(EfficientNetConfig("b5", 13, 4, class_distribution=[0.25] * 4))
    model.load_state_dict(sd)
    model.to(DEV).eval()
    B = 6
    x = detgen.normal("bc.x", (B, 13, 256, 256), seed=41).to(DEV)
    with torch.no_grad():
        full = model(x)
        for i in (0, B - 1):
            one = model(x[i:i + 1])
            assert rel_err(full[i:i + 1].cpu().numpy(), one.cpu().numpy()) < 1e-4, i
    assert torch.isfinite(full).all()


def test_prithvi_seg_eval_batch_independent_past_2gib():
    from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
    from s2lc_amd.modules.prithvi_segmentation import PrithviSegmentationNet, PrithviSegmentationNetConfig

    torch.manual_seed(3)
    bb = MaskedAutoencoderViT(**PRITHVI_FULL, _decoder=False, _flat=False)
    net = PrithviSegmentationNet(PrithviSegmentationNetConfig(1, 4, 256, 1, 0.1, True), backbone=bb).to(DEV).eval()
    B = 15                                   # 15 x 768 x 224 x 224 x 4 B = 2.3 GiB neck output
    x = detgen.normal("bc.seg.x", (B, 6, 1, 224, 224), seed=42).to(DEV)
    noise = detgen.uniform("bc.seg.noise", (B, 196), 0.0, 1.0, seed=42)
    with torch.no_grad():
        net.masking_noise = noise
        full = net(x)
        for i in (0, B - 1):
            net.masking_noise = noise[i:i + 1]
            one = net(x[i:i + 1])
            assert rel_err(full[i:i + 1].cpu().numpy(), one.cpu().numpy()) < 1e-4, i
    assert torch.isfinite(full).all()
