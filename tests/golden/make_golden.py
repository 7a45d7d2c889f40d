"""Generate the golden fixtures under tests/golden/ from the *imported reference*.

Run in the build container only:   python tests/golden/make_golden.py [unet|loss|prithvi|all]

Weights and inputs come from oracle/detgen.py (hash-based, reproducible anywhere), are loaded
into the reference's own nn.Modules with load_state_dict, and the reference's outputs are
stored as strided subsamples + whole-tensor checksums + full uint8 class masks (each file
< ~100 KB).  The fixtures are data only; none of the reference's source travels.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

import ref_harness  # noqa: E402
from oracle import detgen  # noqa: E402
from oracle import efficientnet_unet_ref as R  # noqa: E402


def sub(t: torch.Tensor, n: int = 2048) -> np.ndarray:
    """Deterministic strided subsample of the flattened tensor (<= n values)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).numpy().copy()


def checks(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), d.abs().max().item(), float(d.numel())])


def dc_blocks(model) -> list[int]:
    """Indices of blocks that draw drop-connect noise in training (efficientnet_unet.py:383-385)."""
    out = []
    n = len(model.encoder.blocks)
    for i, b in enumerate(model.encoder.blocks):
        rate = model.encoder.drop_connect_rate * (i / n)
        if b.skip_connection and b.stride == 1 and b.input_filters == b.output_filters and rate:
            out.append(i)
    return out


def build_ref_unet(ns, version: str, C: int, ncls: int, native: bool):
    """native: the unmodified EfficientnetUnet (only valid at 224x224x6).  Otherwise the
    reference's own sub-modules wired per SURVEY §8 a7-G (size[4] = 32 + C; drop head-sized maps)."""
    cfg = ns.unet.EfficientNetConfig(version=version, in_channels=C, num_classes=ncls,
                                     class_distribution=[1.0 / ncls] * ncls)
    if native:
        return ns.unet.EfficientnetUnet(cfg)
    U = ns.unet

    class Wired(torch.nn.Module):
        def __init__(self):
            super().__init__()
            m6 = U.EfficientnetUnet(U.EfficientNetConfig(version=version, in_channels=6, num_classes=ncls,
                                                         class_distribution=[1.0 / ncls] * ncls))
            self.encoder = U.EfficientNet(cfg)
            self.up_convs, self.double_convs = m6.up_convs, m6.double_convs
            self.input_up_conv = m6.input_up_conv
            self.input_double_conv = U._double_conv(32 + C, 32)
            self.out_conv1x1 = m6.out_conv1x1

        def forward(self, x):
            identity = x
            enc = self.encoder
            h = enc.stem(x)
            cands = []
            n = len(enc.blocks)
            for i, blk in enumerate(enc.blocks):
                h = blk(h, drop_connect_rate=enc.drop_connect_rate * (i / n))
                cands.append(h)
            fm = []
            for c in cands:
                if c.shape[-2:] not in [f.shape[-2:] for f in fm] and c.shape[-2:] != h.shape[-2:]:
                    fm.insert(0, c)
            h = enc.conv_head(h)
            for up, dc, f in zip(self.up_convs, self.double_convs, fm):
                h = dc(torch.cat([up(h), f], dim=1))
            h = self.input_double_conv(torch.cat([self.input_up_conv(h), identity], dim=1))
            return self.out_conv1x1(h)

    return Wired()


def unet_case(ns, tag, version, C, H, B, ncls, native, train, seed):
    torch.manual_seed(0)
    model = build_ref_unet(ns, version, C, ncls, native)
    net = R.build(version, C, ncls)
    shapes = R.state_shapes(net)
    sd = detgen.fill_state(shapes, seed=seed)
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    x = detgen.normal(f"{tag}.x", (B, C, H, H), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, H, H), ncls, seed=seed)
    out = {"meta": np.array([C, H, B, ncls, int(native), int(train), seed])}
    nblk = len(net.blocks)
    noise = detgen.uniform(f"{tag}.dc", (nblk, B), 0.0, 1.0, seed=seed)
    if train:
        model.train()
        draws = [noise[i] for i in dc_blocks(model if native else model)]
        with ref_harness.injected_rand(draws):
            logits = model(x)
        alpha = torch.ones(ncls)
        fl = ns.losses.FocalLoss(alpha=alpha, gamma=2.0, label_smoothing=0.0, ignore_index=0)
        ce = torch.nn.CrossEntropyLoss(ignore_index=0)
        loss_f = fl(logits, y)
        loss_c = ce(logits, y)
        out["loss_focal"] = np.array([loss_f.item()])
        out["loss_ce"] = np.array([loss_c.item()])
        loss_f.backward()
        named = dict(model.named_parameters())
        grad_names = ["encoder.stem.0.weight", "encoder.blocks.1.stem.3.weight",
                      "encoder.blocks.1.squeeze_excitation.3.bias", "encoder.blocks.2.final_layer.0.weight",
                      "encoder.blocks.2.final_layer.1.weight", "encoder.conv_head.0.weight",
                      "up_convs.0.weight", "up_convs.3.bias", "double_convs.0.0.weight",
                      "double_convs.2.4.bias", "input_double_conv.0.weight", "out_conv1x1.weight",
                      "out_conv1x1.bias"]
        blk_dw = next(k for k in named if re.match(r"encoder\.blocks\.\d+\.stem\.3\.weight", k))
        grad_names = [g for g in grad_names if g in named] + [blk_dw, blk_dw.replace("stem.3.weight", "stem.0.weight")]
        for gname in dict.fromkeys(grad_names):
            g = named[gname].grad
            out["grad:" + gname] = sub(g, 512)
            out["gradck:" + gname] = checks(g)
        tot = 0.0
        for k, p in named.items():
            if p.grad is not None:
                tot += p.grad.double().pow(2).sum().item()
        out["grad_total_sq"] = np.array([tot])
        out["grad_none"] = np.array([k for k, p in named.items() if p.grad is None])
        new_sd = model.state_dict()
        for bname in ["encoder.stem.1", "encoder.blocks.3.stem.4", "double_convs.1.1"]:
            out["rm:" + bname] = new_sd[bname + ".running_mean"].numpy().copy()
            out["rv:" + bname] = new_sd[bname + ".running_var"].numpy().copy()
            out["nbt:" + bname] = np.array([int(new_sd[bname + ".num_batches_tracked"])])
    else:
        model.eval()
        with torch.no_grad():
            logits = model(x)
        alpha = torch.ones(ncls)
        fl = ns.losses.FocalLoss(alpha=alpha, gamma=2.0, label_smoothing=0.0, ignore_index=0)
        out["loss_focal"] = np.array([fl(logits, y).item()])
        out["loss_ce"] = np.array([torch.nn.CrossEntropyLoss(ignore_index=0)(logits, y).item()])
    out["logits_sub"] = sub(logits, 4096)
    out["logits_ck"] = checks(logits)
    out["mask"] = logits.argmax(dim=1).to(torch.uint8).numpy()
    # margin between best and second-best logit: tells the test where argmax is numerically fragile
    top2 = logits.detach().topk(2, dim=1).values
    out["margin_min"] = np.array([(top2[:, 0] - top2[:, 1]).min().item()])
    np.savez_compressed(HERE / f"unet_{tag}.npz", **out)
    print(f"unet_{tag}: logits ck {out['logits_ck']}, focal {out['loss_focal']}")


def gen_unet(ns):
    # F1: unmodified reference, b0 224x224x6, eval, bs 1 & 2
    unet_case(ns, "b0_224_eval_bs1", "b0", 6, 224, 1, 4, True, False, 1)
    unet_case(ns, "b0_224_eval_bs2", "b0", 6, 224, 2, 4, True, False, 2)
    # F2: train mode with injected drop-connect noise (+ grads, running stats)
    unet_case(ns, "b0_224_train_bs2", "b0", 6, 224, 2, 4, True, True, 3)
    # F3: b5 reference-native shape
    unet_case(ns, "b5_224_eval_bs1", "b5", 6, 224, 1, 4, True, False, 4)
    # F4: BASELINE shapes through the reference's own sub-modules wired per a7-G
    unet_case(ns, "b0_128x4_eval_bs1", "b0", 4, 128, 1, 4, False, False, 5)
    unet_case(ns, "b0_128x4_train_bs2", "b0", 4, 128, 2, 4, False, True, 6)
    unet_case(ns, "b5_256x13_eval_bs1", "b5", 13, 256, 1, 4, False, False, 7)
    unet_case(ns, "b5_64x13_train_bs2", "b5", 13, 64, 2, 4, False, True, 8)


def gen_loss(ns):
    """F6: FocalLoss / CrossEntropy edge cases straight from the reference's classes."""
    out = {}
    B, C, H = 2, 4, 16
    lg = detgen.normal("loss.logits", (B, C, H, H), std=2.0, seed=11)
    y = detgen.labels("loss.y", (B, H, H), C, p_zero=0.2, seed=11)
    out["shape"] = np.array([B, C, H])
    FL = ns.losses.FocalLoss
    cases = {
        "focal_g2": lambda l, t: FL(torch.ones(C), 2.0, 0.0, ignore_index=0)(l, t),
        "focal_g0p5_ls": lambda l, t: FL(torch.ones(C), 0.5, 0.1, ignore_index=0)(l, t),
        "focal_alpha": lambda l, t: FL(torch.tensor([0.1, 0.9, 0.6, 0.7]), 2.0, 0.0, ignore_index=0)(l, t),
        "focal_noignore": lambda l, t: FL(torch.ones(C), 2.0, 0.0, ignore_index=-100)(l, t),
        "focal_sum": lambda l, t: FL(torch.ones(C), 2.0, 0.0, ignore_index=0, reduce_type="sum")(l, t),
        "ce_masked": lambda l, t: torch.nn.CrossEntropyLoss(ignore_index=0)(l, t),
        "ce_plain": lambda l, t: torch.nn.CrossEntropyLoss(ignore_index=-100)(l, t),
        "ce_w_ls": lambda l, t: torch.nn.CrossEntropyLoss(weight=torch.tensor([0.1, 0.9, 0.6, 0.7]),
                                                          label_smoothing=0.1, ignore_index=0)(l, t),
    }
    for name, fn in cases.items():
        l = lg.clone().requires_grad_(True)
        v = fn(l, y)
        v.backward()
        out["val:" + name] = np.array([v.item()])
        out["grad:" + name] = l.grad.numpy().copy()
    # all-ignored batch
    y0 = torch.zeros_like(y)
    l = lg.clone().requires_grad_(True)
    v = FL(torch.ones(C), 2.0, 0.0, ignore_index=0)(l, y0)
    v.backward()
    out["val:focal_allignored"] = np.array([v.item()])
    out["grad:focal_allignored"] = l.grad.numpy().copy()
    v = torch.nn.CrossEntropyLoss(ignore_index=0)(lg, y0)
    out["val:ce_allignored"] = np.array([v.item()])  # NaN in torch
    # 2-class
    lg2 = detgen.normal("loss.logits2", (B, 2, H, H), std=2.0, seed=12)
    y2 = detgen.labels("loss.y2", (B, H, H), 2, p_zero=0.4, seed=12)
    l = lg2.clone().requires_grad_(True)
    v = FL(torch.ones(2), 2.0, 0.0, ignore_index=0)(l, y2)
    v.backward()
    out["val:focal_2class"] = np.array([v.item()])
    out["grad:focal_2class"] = l.grad.numpy().copy()
    # get_loss weights (losses.py:26-29) via the reference function itself
    import types
    cfgm = types.SimpleNamespace(num_classes=4, train=types.SimpleNamespace(
        weighted_loss=True, class_distribution=[0.4, 0.3, 0.2, 0.1], masked_loss=True,
        loss_type=ns.losses.LossType.FOCAL, focal_loss_gamma=2.0, label_smoothing=0.0))
    fl = ns.losses.get_loss(cfgm)
    out["get_loss_alpha_masked"] = fl.alpha.numpy().copy()
    cfgm.train.masked_loss = False
    cfgm.train.class_distribution = [0.4, 0.3, 0.2, 0.1]
    out["get_loss_alpha_unmasked"] = ns.losses.get_loss(cfgm).alpha.numpy().copy()
    np.savez_compressed(HERE / "loss_cases.npz", **out)
    print("loss_cases written")


def gen_ops(ns):
    """F5: per-op micro-fixtures: Conv2dSamePadding, MBConvBlock residual quirk, _drop_connect."""
    U = ns.unet
    out = {}
    for name, (k, s, H, W, groups, cin, cout) in {
        "same_k3s2_even": (3, 2, 16, 16, 1, 3, 5), "same_k3s2_odd": (3, 2, 15, 13, 1, 3, 5),
        "same_k5s2_even": (5, 2, 16, 12, 4, 4, 4), "same_k5s2_odd": (5, 2, 9, 7, 4, 4, 4),
        "same_k5s1": (5, 1, 7, 9, 4, 4, 4), "same_k3s1": (3, 1, 6, 6, 1, 2, 3),
    }.items():
        conv = U.Conv2dSamePadding(cin, cout, k, stride=s, groups=groups, bias=False)
        w = detgen.uniform(name + ".w", tuple(conv.weight.shape), -1, 1)
        conv.weight.data.copy_(w)
        x = detgen.normal(name + ".x", (2, cin, H, W))
        out[name + ":y"] = conv(x).detach().numpy().copy()
        out[name + ":cfg"] = np.array([k, s, H, W, groups, cin, cout])
    x = detgen.normal("dc.x", (4, 3, 2, 2))
    u = torch.tensor([0.05, 0.5, 0.85, 0.95])
    with ref_harness.injected_rand([u]):
        out["dropconnect:y"] = U._drop_connect(x, 0.1, True).numpy().copy()
    # residual quirk: first-of-stage block with in==out and stride (1,1) must NOT add identity
    cfg = U.EfficientNetConfig(version="b0", in_channels=6, num_classes=4, class_distribution=[.25] * 4)
    bc = U.BlockConfig.from_str("r2_k3_s11_e6_i16_o16_se0.25")
    first = U.MBConvBlock(bc, cfg)
    bc.stride = 1
    rep = U.MBConvBlock(bc, cfg)
    rep.load_state_dict(first.state_dict())
    first.eval(); rep.eval()
    x = detgen.normal("quirk.x", (1, 16, 8, 8))
    with torch.no_grad():
        d = (rep(x) - first(x) - x).abs().max().item()
    out["quirk:first_has_no_residual"] = np.array([d])
    np.savez_compressed(HERE / "ops_cases.npz", **out)
    print("ops_cases written; quirk residual delta", d)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ns = ref_harness.load()
    torch.set_num_threads(8)
    if what in ("unet", "all"):
        gen_unet(ns)
    if what in ("loss", "all"):
        gen_loss(ns)
    if what in ("ops", "all"):
        gen_ops(ns)
    if what in ("prithvi", "all"):
        try:
            from make_golden_prithvi import gen_prithvi
        except ImportError:
            gen_prithvi = None
        if gen_prithvi:
            gen_prithvi(ns)
