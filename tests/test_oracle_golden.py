"""CPU: the oracle restatement reproduces the imported reference's outputs (golden fixtures)."""
import numpy as np
import pytest
import torch

from oracle import detgen, losses_ref
from oracle import efficientnet_unet_ref as R
from tests.helpers import UNET_CASES, checks, load, rel_err, sub


def run_oracle_case(tag):
    version, C, H, B, ncls, train, seed = UNET_CASES[tag]
    net = R.build(version, C, ncls)
    sd = detgen.fill_state(R.state_shapes(net), seed=seed)
    x = detgen.normal(f"{tag}.x", (B, C, H, H), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, H, H), ncls, seed=seed)
    noise = detgen.uniform(f"{tag}.dc", (len(net.blocks), B), 0.0, 1.0, seed=seed)
    return net, sd, x, y, noise


@pytest.mark.parametrize("tag", [t for t, v in UNET_CASES.items() if not v[5]])
def test_unet_eval_matches_reference(tag):
    g = load(f"unet_{tag}.npz")
    net, sd, x, y, _ = run_oracle_case(tag)
    with torch.no_grad():
        logits = R.unet_forward(sd, net, x, training=False)
    assert rel_err(sub(logits, 4096), g["logits_sub"]) < 1e-5
    assert abs(checks(logits)[1] - g["logits_ck"][1]) / g["logits_ck"][1] < 1e-5
    mask = losses_ref.class_mask(logits).to(torch.uint8).numpy()
    assert np.array_equal(mask, g["mask"])  # class masks bit-exact
    fl = losses_ref.focal(logits, y, torch.ones(net.num_classes), 2.0, 0.0, ignore_index=0)
    ce = losses_ref.cross_entropy(logits, y, ignore_index=0)
    assert abs(fl.item() - g["loss_focal"][0]) < 1e-5 * abs(g["loss_focal"][0])
    assert abs(ce.item() - g["loss_ce"][0]) < 1e-5 * abs(g["loss_ce"][0])


@pytest.mark.parametrize("tag", [t for t, v in UNET_CASES.items() if v[5]])
def test_unet_train_matches_reference(tag):
    g = load(f"unet_{tag}.npz")
    net, sd, x, y, noise = run_oracle_case(tag)
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    newbuf = {}
    logits = R.unet_forward(sd, net, x, training=True, dc_noise=noise, new_buffers=newbuf)
    assert rel_err(sub(logits, 4096), g["logits_sub"]) < 2e-5
    assert np.array_equal(losses_ref.class_mask(logits).to(torch.uint8).numpy(), g["mask"])
    fl = losses_ref.focal(logits, y, torch.ones(net.num_classes), 2.0, 0.0, ignore_index=0)
    assert abs(fl.item() - g["loss_focal"][0]) < 1e-5 * abs(g["loss_focal"][0])
    fl.backward()
    none = set(g["grad_none"].tolist())
    tot = 0.0
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None:
            tot += v.grad.double().pow(2).sum().item()
    assert abs(tot - g["grad_total_sq"][0]) < 1e-3 * g["grad_total_sq"][0]
    for key in g.files:
        if key.startswith("grad:"):
            name = key[5:]
            assert rel_err(sub(sd[name].grad, 512), g[key]) < 2e-3, name
    assert {"encoder.fc.3.weight", "encoder.fc.3.bias"} <= none
    for key in g.files:
        if key.startswith("rm:"):
            assert rel_err(newbuf[key[3:] + ".running_mean"].numpy(), g[key]) < 1e-5
        if key.startswith("rv:"):
            assert rel_err(newbuf[key[3:] + ".running_var"].numpy(), g[key]) < 1e-5
        if key.startswith("nbt:"):
            assert int(newbuf[key[4:] + ".num_batches_tracked"]) == int(g[key][0])


def test_loss_cases_match_reference():
    g = load("loss_cases.npz")
    B, C, H = g["shape"]
    lg = detgen.normal("loss.logits", (B, C, H, H), std=2.0, seed=11)
    y = detgen.labels("loss.y", (B, H, H), C, p_zero=0.2, seed=11)
    ones = torch.ones(C)
    aw = torch.tensor([0.1, 0.9, 0.6, 0.7])
    cases = {
        "focal_g2": lambda l, t: losses_ref.focal(l, t, ones, 2.0, 0.0, 0),
        "focal_g0p5_ls": lambda l, t: losses_ref.focal(l, t, ones, 0.5, 0.1, 0),
        "focal_alpha": lambda l, t: losses_ref.focal(l, t, aw, 2.0, 0.0, 0),
        "focal_noignore": lambda l, t: losses_ref.focal(l, t, ones, 2.0, 0.0, -100),
        "focal_sum": lambda l, t: losses_ref.focal(l, t, ones, 2.0, 0.0, 0, "sum"),
        "ce_masked": lambda l, t: losses_ref.cross_entropy(l, t, None, 0.0, 0),
        "ce_plain": lambda l, t: losses_ref.cross_entropy(l, t, None, 0.0, -100),
        "ce_w_ls": lambda l, t: losses_ref.cross_entropy(l, t, aw, 0.1, 0),
    }
    for name, fn in cases.items():
        l = lg.clone().requires_grad_(True)
        v = fn(l, y)
        v.backward()
        assert abs(v.item() - g["val:" + name][0]) <= 2e-6 * abs(g["val:" + name][0]), name
        assert rel_err(l.grad.numpy(), g["grad:" + name]) < 1e-4, name
    y0 = torch.zeros_like(y)
    l = lg.clone().requires_grad_(True)
    v = losses_ref.focal(l, y0, ones, 2.0, 0.0, 0)
    v.backward()
    assert v.item() == 0.0 == g["val:focal_allignored"][0]
    assert np.abs(l.grad.numpy()).max() == 0.0 and np.abs(g["grad:focal_allignored"]).max() == 0.0
    assert np.isnan(losses_ref.cross_entropy(lg, y0, None, 0.0, 0).item()) and np.isnan(g["val:ce_allignored"][0])
    lg2 = detgen.normal("loss.logits2", (B, 2, H, H), std=2.0, seed=12)
    y2 = detgen.labels("loss.y2", (B, H, H), 2, p_zero=0.4, seed=12)
    l = lg2.clone().requires_grad_(True)
    v = losses_ref.focal(l, y2, torch.ones(2), 2.0, 0.0, 0)
    v.backward()
    assert abs(v.item() - g["val:focal_2class"][0]) < 2e-6
    assert rel_err(l.grad.numpy(), g["grad:focal_2class"]) < 1e-4
    assert np.allclose(losses_ref.loss_class_weights([0.4, 0.3, 0.2, 0.1], True).numpy(), g["get_loss_alpha_masked"])
    assert np.allclose(losses_ref.loss_class_weights([0.4, 0.3, 0.2, 0.1], False).numpy(), g["get_loss_alpha_unmasked"])


def test_op_micro_cases_match_reference():
    import torch.nn.functional as F

    g = load("ops_cases.npz")
    for name in ["same_k3s2_even", "same_k3s2_odd", "same_k5s2_even", "same_k5s2_odd", "same_k5s1", "same_k3s1"]:
        k, s, H, W, groups, cin, cout = g[name + ":cfg"]
        w = detgen.uniform(name + ".w", (int(cout), int(cin // groups), int(k), int(k)), -1, 1)
        x = detgen.normal(name + ".x", (2, int(cin), int(H), int(W)))
        y = F.conv2d(R.same_pad(x, int(k), int(s)), w, None, int(s), 0, 1, int(groups))
        assert rel_err(y.numpy(), g[name + ":y"]) < 1e-6, name
    x = detgen.normal("dc.x", (4, 3, 2, 2))
    u = torch.tensor([0.05, 0.5, 0.85, 0.95])
    assert np.array_equal(R.drop_connect(x, 0.1, u).numpy(), g["dropconnect:y"])
    assert g["quirk:first_has_no_residual"][0] < 1e-5  # reference: first-of-stage never adds identity
    assert not R.RefBlock(3, 1, 16, 16, 6, 4, True).residual and R.RefBlock(3, 1, 16, 16, 6, 4, False).residual


def test_b5_block_table_matches_survey():
    net = R.build("b5", 13, 4)
    assert net.stem_out == 48 and len(net.blocks) == 39 and net.head_out == 2048
    outs = []
    for b in net.blocks:
        if b.first_of_stage:
            outs.append(b.cout)
    assert outs == [24, 40, 64, 128, 176, 304, 512]
