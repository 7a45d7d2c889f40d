"""CPU: the PRODUCT's host-side functions (not the oracle's) against fixtures generated from the imported reference:
get_3d_sincos_pos_embed, patchify / unpatchify (the reference's trainer calls unpatchify, train_mae_prithvi.py:182),
get_loss(config) (losses.py:24-63), the CNES / OSM label look-up tables (configs/cnes_labell_mappings.py:78-95), the
checkpoint loaders (utils.py:62-96; `net._orig_mod.` prefix of Lightning + torch.compile checkpoints)."""
import types

import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import detgen
from tests.helpers import PRITHVI_SMALL, PRITHVI_SMALL_T3, checks, load, sub


def test_product_sincos_tables_equal_reference():
    from s2lc_amd.modules.prithvi import get_3d_sincos_pos_embed

    g = load("prithvi_misc.npz")
    for dim, grid in ((768, (1, 14, 14)), (512, (1, 14, 14)), (512, (3, 14, 14)), (32, (3, 2, 2)), (16, (1, 4, 4))):
        t = torch.from_numpy(get_3d_sincos_pos_embed(dim, grid, cls_token=True)).float()
        key = f"pos:{dim}:{grid[0]}x{grid[1]}x{grid[2]}"
        assert np.array_equal(sub(t, 2048), g[key + ":sub"]), key          # bit-identical
        assert np.allclose(checks(t), g[key + ":ck"], rtol=1e-12), key


def test_product_patchify_unpatchify_equal_reference():
    from s2lc_amd.modules.prithvi import MaskedAutoencoderViT

    g = load("prithvi_misc.npz")
    for tag, args in {"s1": PRITHVI_SMALL, "s3": PRITHVI_SMALL_T3}.items():
        m = MaskedAutoencoderViT(**args)
        s = m.spec
        x = detgen.normal(f"pf.{tag}", (2, s.in_chans, s.num_frames, s.img_size, s.img_size), seed=6)
        pt = m.patchify(x)
        assert np.array_equal(sub(pt, 1024), g[f"pf:{tag}:sub"])
        assert np.allclose(checks(pt), g[f"pf:{tag}:ck"], rtol=1e-12)
        assert torch.equal(m.unpatchify(pt), x)


def _cfg(**train):
    from s2lc_amd.losses import LossType

    base = dict(weighted_loss=True, class_distribution=[0.4, 0.3, 0.2, 0.1], masked_loss=True, loss_type=LossType.FOCAL,
                focal_loss_gamma=2.0, label_smoothing=0.0)
    base.update(train)
    return types.SimpleNamespace(num_classes=4, train=types.SimpleNamespace(**base))


def test_product_get_loss_equals_reference_factory():
    from s2lc_amd.losses import CrossEntropyLoss, FocalLoss, LossType, get_loss

    g = load("loss_cases.npz")
    fl = get_loss(_cfg())
    assert isinstance(fl, FocalLoss) and fl.ignore_index == 0 and fl.gamma == 2.0
    assert np.array_equal(fl.alpha.numpy(), g["get_loss_alpha_masked"])
    fl = get_loss(_cfg(masked_loss=False))
    assert fl.ignore_index == -100 and np.array_equal(fl.alpha.numpy(), g["get_loss_alpha_unmasked"])
    fl = get_loss(_cfg(weighted_loss=False))
    assert torch.equal(fl.alpha, torch.ones(4))
    ce = get_loss(_cfg(loss_type=LossType.CE, label_smoothing=0.1))
    assert isinstance(ce, CrossEntropyLoss) and ce.ignore_index == 0 and ce.label_smoothing == 0.1
    assert np.array_equal(ce.weight.numpy(), g["get_loss_alpha_masked"])
    with pytest.raises(ValueError):
        get_loss(_cfg(loss_type="nope"))
    with pytest.raises(NotImplementedError):
        get_loss(_cfg(loss_type=LossType.DICE))


def test_label_luts_equal_reference_transform():
    """`label_lut(name)` against the reference's get_cnes_transform applied to every uint8 value, for EVERY map name the
    reference registers (configs/data_config.py:80-90)."""
    from s2lc_amd.data.gpu_pipeline import label_lut

    g = load("label_luts.npz")
    names = [k[4:] for k in g.files if k.startswith("lut:")]
    assert len(names) == 9
    for name in names:
        assert np.array_equal(label_lut(name).numpy().astype(np.int64), g["lut:" + name]), name


def test_load_reference_checkpoint_accepts_lightning_compile_prefix(tmp_path):
    """Lightning saves `state_dict` of the LightningModule whose `net` is a torch.compile wrapper: keys are
    `net._orig_mod.<name>` (train_segmentation.py:70-75,247-255)."""
    from oracle import efficientnet_unet_ref as R
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
    from s2lc_amd.utils import load_reference_checkpoint

    net = R.build("b0", 4, 4)
    sd = detgen.fill_state(R.state_shapes(net), seed=9)
    for wrap in ("net._orig_mod.", "net.", "_orig_mod.", ""):
        ck = {"state_dict": {wrap + k: v for k, v in sd.items()}, "hyper_parameters": {}}
        path = tmp_path / "last.ckpt"
        torch.save(ck, path)
        model = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4))
        load_reference_checkpoint(model, path)
        got = model.state_dict()
        assert list(got) == list(sd)
        assert all(torch.equal(got[k], sd[k]) for k in sd)
    torch.save({wrap + k: v for k, v in sd.items() if "stem.0" not in k}, path)     # bare state dict with a missing key
    with pytest.raises(RuntimeError, match="Missing key"):
        load_reference_checkpoint(EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4)), path)


def test_load_prithvi_pop_and_reinit_rules(tmp_path):
    """utils.py:62-96: pos_embed / decoder_pos_embed are popped from the checkpoint and re-initialised for `num_frames`; with
    no_decoder the decoder entries are dropped and the module has no decoder parameters; everything else is loaded."""
    from s2lc_amd.modules.prithvi import MaskedAutoencoderViT, get_3d_sincos_pos_embed
    from s2lc_amd.utils import _prithvi_model_args, load_prithvi

    args = _prithvi_model_args(3)                      # the published checkpoint was trained with 3 frames
    torch.manual_seed(1)
    src = MaskedAutoencoderViT(**args, _flat=False)
    ck = {k: torch.randn_like(v) * 0.02 for k, v in src.state_dict().items()}
    ck["pos_embed"] = torch.full_like(ck["pos_embed"], 7.0)              # must NOT survive the load
    ck["decoder_pos_embed"] = torch.full_like(ck["decoder_pos_embed"], 7.0)
    path = tmp_path / "Prithvi_100M.pt"
    torch.save(ck, path)
    m = load_prithvi(num_frames=1, no_decoder=True, weights=str(path))
    sd = m.state_dict()
    assert not any(k.startswith(("decoder_embed", "mask_token", "decoder_blocks", "decoder_norm", "decoder_pred")) for k in sd)
    assert "decoder_pos_embed" in sd                      # the table itself stays a (re-initialised) parameter, as in the reference
    want = torch.from_numpy(get_3d_sincos_pos_embed(768, (1, 14, 14), cls_token=True)).float().unsqueeze(0)
    assert torch.equal(sd["pos_embed"], want) and sd["pos_embed"].shape == (1, 197, 768)
    for k in ("cls_token", "patch_embed.proj.bias", "blocks.3.attn.qkv.weight", "blocks.11.mlp.fc2.bias", "norm.weight"):
        assert torch.equal(sd[k], ck[k]), k
    assert not m.pos_embed.requires_grad
    full = load_prithvi(num_frames=3, no_decoder=False, weights=str(path))
    sdf = full.state_dict()
    assert torch.equal(sdf["decoder_pred.weight"], ck["decoder_pred.weight"]) and torch.equal(sdf["mask_token"], ck["mask_token"])
    assert sdf["pos_embed"].shape == (1, 3 * 196 + 1, 768) and not torch.equal(sdf["pos_embed"], ck["pos_embed"])
    assert list(sdf) == list(src.state_dict())            # registration order of the reference kept
    with pytest.raises(FileNotFoundError):
        load_prithvi(1, weights=str(tmp_path / "missing.pt"))


def test_standalone_efficientnet_accepts_foreign_state_dict_like_the_reference_self_test():
    """reference efficientnet_unet.py:415-431 (`_test`): every version b0-b7 constructs as a standalone `EfficientNet`, and
    `load_state_dict(imagenet_weights, strict=False)` - whose keys follow another naming scheme - neither raises nor touches a
    parameter; the module keeps the reference's own key set."""
    import torch

    from s2lc_amd.modules.efficientnet_unet import EfficientNet, EfficientNetConfig

    for version in ("b0", "b3", "b7"):
        m = EfficientNet(EfficientNetConfig(version, 6, 4, class_distribution=[0.25] * 4))
        keys = list(m.state_dict())
        assert keys[0] == "stem.0.weight" and keys[-1] == "fc.3.bias" and not any(k.startswith("encoder.") for k in keys)
        before = m._flat_params.clone()
        foreign = {"_conv_stem.weight": torch.zeros(32, 3, 3, 3), "_bn0.weight": torch.ones(32), "_fc.bias": torch.zeros(1000)}
        res = m.load_state_dict(foreign, strict=False)
        assert set(res.unexpected_keys) == set(foreign)
        assert set(res.missing_keys) >= {k for k in keys if not k.endswith("num_batches_tracked")}
        assert torch.equal(m._flat_params, before)
        own = {k: torch.full_like(v, 0.5) if v.dtype.is_floating_point else v for k, v in m.state_dict().items()}
        m.load_state_dict(own)
        assert float(m.fc[3].weight.min()) == 0.5 and float(m._flat_params[: m.stem[0].weight.numel()].max()) == 0.5


def test_classification_bias_prior_equals_reference():
    """`initialize_classification_layer_bias` (reference utils.py:174-188) against the reference's own output
    (tests/golden/make_golden_r4.py): the two-class log(p1 / p0) fill, non-uniform log priors on Conv2d and Linear, and the
    reference's error behaviour - its sum check runs on the eps-shifted float32 distribution, so it refuses a valid 10-class
    ramp and an unnormalised one alike; the drop-in refuses exactly the same inputs and leaves the bias untouched."""
    from torch import nn

    from s2lc_amd.utils import initialize_classification_layer_bias

    g = load("class_bias_prior.npz")
    names = sorted({k.split(".")[0] for k in g.files})
    assert {"two_class_conv2", "nonuniform4_conv", "ramp10_conv", "not_normalised"} <= set(names)
    for name in names:
        dist = [float(v) for v in g[f"{name}.dist"]]
        is_linear, n = (int(v) for v in g[f"{name}.kind"])
        torch.manual_seed(0)
        layer = nn.Linear(16, n) if is_linear else nn.Conv2d({2: 8, 1: 8, 3: 8}.get(n, 32), n, 1)
        before = layer.bias.detach().clone()
        if int(g[f"{name}.raised"][0]):
            with pytest.raises(AssertionError):
                initialize_classification_layer_bias(layer, dist)
            assert torch.equal(layer.bias.detach(), before), name
        else:
            initialize_classification_layer_bias(layer, dist)
            assert np.array_equal(layer.bias.detach().numpy(), g[f"{name}.bias"]), name       # bit-identical
        assert layer.bias.requires_grad


def test_unet_classifier_bias_follows_the_class_distribution():
    """the model constructor applies the prior to out_conv1x1 (reference efficientnet_unet.py:131-134)"""
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet

    g = load("class_bias_prior.npz")
    m = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.5, 0.3, 0.15, 0.05]))
    assert np.array_equal(m.out_conv1x1.bias.detach().cpu().numpy(), g["nonuniform4_conv.bias"])


def test_copied_module_gets_its_own_compile_handle():
    """`_compile_handle` is id-based; a deepcopy (or unpickled copy) must register itself, or its compiled forward would
    run the ORIGINAL module's weights (or fail once the original is gone)."""
    import copy

    from s2lc_amd import compile_ops
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet

    m = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4))
    c = copy.deepcopy(m)
    assert c._compile_handle != m._compile_handle
    assert compile_ops._module(c._compile_handle) is c and compile_ops._module(m._compile_handle) is m
    assert c._engines == {} and c._flat_params.data_ptr() != m._flat_params.data_ptr()
    sd_m, sd_c = m.state_dict(), c.state_dict()
    assert list(sd_m) == list(sd_c) and all(torch.equal(sd_m[k], sd_c[k]) for k in sd_m)
