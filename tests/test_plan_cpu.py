"""CPU: the planner's forward AND hand-derived backward programs, run through the op emulator,
reproduce oracle autograd (logits, gradient w.r.t. every parameter, running statistics).

Two modes.  `wide` holds everything in float64 and must agree with a float64 oracle to ~1e-9:
that checks the planner's algebra exactly.  The fp32 mode is compared with the *same float64
truth* and must not be worse than a small multiple of the fp32 oracle's own distance from it:
with train-mode BatchNorm on small maps and ReLU masks, fp32 gradients of this network carry
~1e-2 relative noise in ANY implementation (the reference's included), so a fixed 1e-3 bar on
gradients would be meaningless; the 1e-3 bar of BASELINE.json applies to logits.
"""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import detgen, losses_ref
from oracle import efficientnet_unet_ref as R
from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
from tests.helpers import rel_err
from tests.plan_harness import emulate, fview, make_bases


def _setup(version, C, H, B, ncls, seed, dcr=0.2):
    cfg = EfficientNetConfig(version, C, ncls, class_distribution=[1.0 / ncls] * ncls, drop_connect_rate=dcr)
    model = EfficientnetUnet(cfg)
    net = R.build(version, C, ncls, drop_connect_rate=dcr)
    sd = detgen.fill_state(R.state_shapes(net), seed=seed)
    model.load_state_dict(sd)
    x = detgen.normal("plan.x", (B, C, H, H), seed=seed)
    y = detgen.labels("plan.y", (B, H, H), ncls, seed=seed)
    noise = detgen.uniform("plan.dc", (len(net.blocks), B), 0.0, 1.0, seed=seed)
    return model, net, sd, x, y, noise


def oracle_grads(net, sd, x, y, noise, ncls, dtype):
    sdd = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.detach().to(dtype)
            if not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        sdd[k] = v
    newbuf = {}
    logits = R.unet_forward(sdd, net, x.to(dtype), training=True, dc_noise=noise.to(dtype), new_buffers=newbuf)
    loss = losses_ref.focal(logits, y, torch.ones(ncls, dtype=dtype), 2.0, 0.0, ignore_index=0)
    (dlogits,) = torch.autograd.grad(loss, logits, retain_graph=True)
    loss.backward()
    return sdd, logits.detach(), dlogits, newbuf


# wide cases use drop-connect rates whose keep probabilities are exact in fp32 (the stage record
# carries KEEP as a float), so the float64 comparison is not limited by that 1e-8 rounding
@pytest.mark.parametrize("version,C,H,B,wide,dcr,defer_all", [("b0", 6, 64, 2, True, 0.25, False), ("b5", 13, 64, 2, True, None, False),
                                                               ("b0", 4, 128, 2, False, 0.2, False), ("b0", 6, 64, 2, True, 0.25, True)])
def test_train_programs_match_oracle_autograd(version, C, H, B, wide, dcr, defer_all, monkeypatch):
    """defer_all: every decoder weight gradient is moved into the encoder's backward section (at the benchmark's size the
    planner does that for the large ones): the re-ordered program must give the same gradients and bucket segments that
    still cover the whole gradient buffer."""
    ncls = 4
    if defer_all:
        monkeypatch.setenv("S2K_TUNING", "1")          # planner switches are only honoured together with this one
        monkeypatch.setenv("S2K_DEFER_MIN_GFLOP", "0")
    model, net, sd, x, y, noise = _setup(version, C, H, B, ncls, seed=21, dcr=dcr)
    plan = model._make_plan(B, H, H, True)
    if defer_all:
        kinds = [k for k, _ in plan.bwd.ops]
        first_enc = next(i for i, k in enumerate(kinds) if k in ("DWCONV_DGRAD", "SE_BWD_REDUCE", "SE_BN_SUMS", "SE_FC_BWD"))
        assert kinds[:first_enc].count("WGRAD") <= 1 and kinds.count("WGRAD") > 20   # (the first deferred one sits at the boundary)
        segs = plan.bwd_param_marks
        assert segs[0][3] == plan.layout.n_params and segs[-1][2] == 0 and all(a[2] == b[3] for a, b in zip(segs, segs[1:]))
    bases = make_bases(plan, model._flat_params, model._flat_bufs, x, noise, B * ncls * H * H, wide)
    emulate(plan.fwd.pack(), bases, wide)
    logits = fview(bases, "OUT", wide).view(B, ncls, H, H).clone()

    sd64, logits64, dlogits64, newbuf64 = oracle_grads(net, sd, x, y, noise, ncls, torch.float64)
    if wide:
        assert rel_err(logits.numpy(), logits64.numpy()) < 1e-6
        fview(bases, "DOUT", wide).copy_(dlogits64.reshape(-1))
    else:
        sd32, logits32, dlogits32, _ = oracle_grads(net, sd, x, y, noise, ncls, torch.float32)
        assert rel_err(logits.numpy(), logits64.numpy()) < 1e-4  # logits: well inside the 1e-3 bar
        fview(bases, "DOUT", wide).copy_(dlogits32.reshape(-1))
    emulate(plan.bwd.pack(), bases, wide)
    grads = fview(bases, "GRADS", wide)
    scale = max(v.grad.abs().max().item() for v in sd64.values() if v.requires_grad and v.grad is not None)
    errs_em, errs_or = [], []
    for name, (off, shape) in plan.layout.params.items():
        g = grads[off:off + int(np.prod(shape))].view(shape)
        ref = sd64[name].grad
        if ref is None:
            assert name.startswith("encoder.fc.") and g.abs().max() == 0
            continue
        if ref.abs().max().item() < 1e-9 * scale:
            # analytically zero (a per-channel constant in front of train-mode BN): we emit exact 0
            assert g.abs().max().item() <= 1e-6 * scale, name
            continue
        e = rel_err(g.numpy(), ref.numpy())
        if wide:
            assert e < 1e-5, (name, e)
        else:
            errs_em.append(e)
            errs_or.append(rel_err(sd32[name].grad.numpy(), ref.numpy()))
    if not wide:
        errs_em, errs_or = np.array(errs_em), np.array(errs_or)
        assert np.median(errs_em) < 3 * np.median(errs_or) + 1e-4
        assert errs_em.max() < 5 * errs_or.max() + 1e-3
    bufs = fview(bases, "BUFS", wide)
    for name, (off, shape) in plan.layout.bufs.items():
        got = bufs[off:off + shape[0]]
        assert rel_err(got.numpy(), newbuf64[name].numpy()) < (1e-6 if wide else 1e-5), name


def test_eval_program_matches_oracle():
    model, net, sd, x, y, noise = _setup("b0", 4, 64, 1, 4, seed=22)
    model.eval()
    plan = model._make_plan(1, 64, 64, False)
    assert plan.bwd is None
    bases = make_bases(plan, model._flat_params, model._flat_bufs, x, noise, 4 * 64 * 64)
    from s2lc_amd.plan import opdefs as D

    before = bases[D.BASE["BUFS"]].clone()
    emulate(plan.fwd.pack(), bases)
    logits = fview(bases, "OUT").view(1, 4, 64, 64)
    with torch.no_grad():
        ref = R.unet_forward(sd, net, x, training=False)
    assert rel_err(logits.numpy(), ref.numpy()) < 1e-5
    assert torch.equal(before, bases[D.BASE["BUFS"]])  # eval never touches running stats


def test_state_dict_surface_matches_reference_names():
    for version, C in (("b0", 6), ("b3", 4), ("b5", 13)):
        model = EfficientnetUnet(EfficientNetConfig(version, C, 4, class_distribution=[.25] * 4))
        shapes = R.state_shapes(R.build(version, C, 4))
        sd = model.state_dict()
        assert list(sd) == list(shapes)
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    # classifier bias = log prior (utils.py:174-188)
    m = EfficientnetUnet(EfficientNetConfig("b0", 6, 4, class_distribution=[.1, .2, .3, .4]))
    assert torch.allclose(m.out_conv1x1.bias, (torch.tensor([.1, .2, .3, .4]) + 1e-6).log())
    with pytest.raises(ValueError):
        EfficientNetConfig("b9", 6, 4)


def test_forward_without_gpu_fails_loudly():
    m = EfficientnetUnet(EfficientNetConfig("b0", 6, 4, class_distribution=[.25] * 4))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 6, 64, 64))


def test_reducer_plans_keep_gradient_buckets_progressive():
    """At the benchmark size the single-GPU plan defers the decoder's weight gradients, so every bucket is final only at the
    end of the backward; a plan built for the data-parallel reducer (defer_wgrads=False, what FlatGradReducer sets) must
    finish its first buckets early so their all-reduce overlaps the rest of the backward.  Planning only, nothing is run."""
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
    from s2lc_amd.plan.unet_plan import plan_unet

    model = EfficientnetUnet(EfficientNetConfig("b5", 13, 4, class_distribution=[0.25] * 4))
    single = plan_unet(model.spec, 32, 256, 256, True, model._layout)
    ddp = plan_unet(model.spec, 32, 256, 256, True, model._layout, defer_wgrads=False)
    n_ops = len(ddp.bwd.ops)
    assert len(ddp.bwd_param_marks) >= 3
    first_end = [seg[1] for seg in ddp.bwd_param_marks]
    assert first_end[0] < 0.35 * n_ops and first_end[1] < 0.8 * n_ops, first_end
    assert single.bwd_param_marks[0][1] > 0.9 * len(single.bwd.ops)      # deferred: the decoder bucket closes at the very end
    covered = sorted((lo, hi) for _, _, lo, hi in ddp.bwd_param_marks)
    assert covered[0][0] == 0 and covered[-1][1] == model._layout.n_params and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_eval_mode_backward_program_matches_oracle_autograd():
    """torch differentiates an eval()-mode module too (BatchNorm on its running statistics, no drop-connect): the eval plan
    built with want_bwd carries a backward program whose BatchNorm backward has no batch-statistics terms.  float64, exact."""
    ncls, B, H = 4, 2, 64
    model, net, sd, x, y, noise = _setup("b0", 6, H, B, ncls, seed=23)
    plan = model._make_plan(B, H, H, False, True)
    assert plan.bwd is not None and not plan.training
    bases = make_bases(plan, model._flat_params, model._flat_bufs, x, noise, B * ncls * H * H, True)
    from s2lc_amd.plan import opdefs as D

    bufs_before = bases[D.BASE["BUFS"]].clone()
    emulate(plan.fwd.pack(), bases, True)
    logits = fview(bases, "OUT", True).view(B, ncls, H, H).clone()
    sdd = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.detach().double()
            if not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        sdd[k] = v
    ref = R.unet_forward(sdd, net, x.double(), training=False)
    loss = losses_ref.focal(ref, y, torch.ones(ncls, dtype=torch.float64), 2.0, 0.0, ignore_index=0)
    (dlogits,) = torch.autograd.grad(loss, ref, retain_graph=True)
    loss.backward()
    assert rel_err(logits.numpy(), ref.detach().numpy()) < 1e-9
    fview(bases, "DOUT", True).copy_(dlogits.reshape(-1))
    emulate(plan.bwd.pack(), bases, True)
    assert torch.equal(bufs_before, bases[D.BASE["BUFS"]])      # eval never touches the running statistics
    grads = fview(bases, "GRADS", True)
    scale = max(v.grad.abs().max().item() for v in sdd.values() if v.requires_grad and v.grad is not None)
    n = 0
    for name, (off, shape) in plan.layout.params.items():
        g = grads[off:off + int(np.prod(shape))].view(shape)
        r = sdd[name].grad
        if r is None:
            assert name.startswith("encoder.fc.") and g.abs().max() == 0
            continue
        # (in eval mode a conv bias in front of BatchNorm DOES get a gradient: the planner must not drop it)
        assert (g - r).abs().max().item() <= 1e-7 * max(r.abs().max().item(), 1e-6 * scale), name
        n += 1
    assert n > 200


@pytest.mark.parametrize("train", [True, False])
def test_input_gradient_program_matches_oracle_autograd(train):
    """want_dx plans (x.requires_grad in torch terms): the gradient w.r.t. the network input — through the stride-2 stem
    (zero-insertion + flipped stride-1 conv) and through the raw-input concat of input_double_conv — float64, exact."""
    ncls, B, H, C = 4, 2, 64, 5
    model, net, sd, x, y, noise = _setup("b0", C, H, B, ncls, seed=29, dcr=0.25)
    plan = plan_unet_for(model, B, H, train, want_dx=True)
    kinds = [k for k, _ in plan.bwd.ops]
    assert kinds.count("UPSAMPLE_ZERO") == 1
    bases = make_bases(plan, model._flat_params, model._flat_bufs, x, noise, B * ncls * H * H, True)
    emulate(plan.fwd.pack(), bases, True)
    sdd = {}
    for k, v in sd.items():
        sdd[k] = v.detach().double() if v.dtype.is_floating_point else v
    xg = x.double().requires_grad_(True)
    logits = R.unet_forward(sdd, net, xg, training=train, dc_noise=noise.double() if train else None, new_buffers={})
    loss = losses_ref.focal(logits, y, torch.ones(ncls, dtype=torch.float64), 2.0, 0.0, ignore_index=0)
    (dlogits,) = torch.autograd.grad(loss, logits, retain_graph=True)
    loss.backward()
    fview(bases, "DOUT", True).copy_(dlogits.reshape(-1))
    emulate(plan.bwd.pack(), bases, True)
    dx = fview(bases, "DX", True).view(B, C, H, H)
    assert torch.isfinite(dx).all()
    assert (dx - xg.grad).abs().max().item() <= 1e-7 * xg.grad.abs().max().item()


def plan_unet_for(model, B, H, train, want_dx=False):
    from s2lc_amd.plan.unet_plan import plan_unet

    return plan_unet(model.spec, B, H, H, train, model._layout, want_bwd=True, want_dx=want_dx)


@pytest.mark.parametrize("train,nested", [(True, False), (False, True)])
def test_efficientnet_encode_and_classifier_programs_float64(train, nested):
    """EfficientNet.encode / .forward as separately callable programs (reference :246-263): outputs, gradients w.r.t. every
    encoder parameter, the input and (classifier) fc, against float64 oracle autograd; standalone layout and the U-Net's."""
    import torch.nn.functional as F

    from oracle.ops_ref import Mem
    from s2lc_amd.plan import opdefs as D
    from s2lc_amd.plan.encoder_plan import plan_encoder
    from s2lc_amd.modules.efficientnet_unet import unet_spec
    from s2lc_amd.plan.unet_plan import build_encoder_layout, build_layout
    from tests.plan_harness import _bytes, flat_from_state

    B, C, H, ncls, p_drop = 2, 5, 64, 3, 0.25
    cfg = EfficientNetConfig("b0", C, ncls, class_distribution=[1.0 / ncls] * ncls, drop_connect_rate=0.25, dropout_rate=p_drop)
    spec = unet_spec(cfg)
    net = R.build("b0", C, ncls, drop_connect_rate=0.25)
    sd = detgen.fill_state(R.state_shapes(net), seed=5)
    sd["encoder.fc.3.weight"] = detgen.normal("fc.w", (ncls, spec.head_out), seed=5) * 0.05
    sd["encoder.fc.3.bias"] = detgen.normal("fc.b", (ncls,), seed=5) * 0.05
    pre = "encoder." if nested else ""
    layout = build_layout(spec) if nested else build_encoder_layout(spec)
    own = {k: v for k, v in sd.items()} if nested else {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    fp, fb = flat_from_state(layout, own)
    x = detgen.normal("enc.x", (B, C, H, H), seed=5)
    dc = detgen.uniform("enc.dc", (len(net.blocks), B), 0.0, 1.0, seed=5)
    du = detgen.uniform("enc.du", (B, spec.head_out), 0.0, 1.0, seed=6)

    def oracle(classifier):
        sdd = {k: (v.detach().double().requires_grad_(not k.endswith(("running_mean", "running_var"))) if v.dtype.is_floating_point else v)
               for k, v in sd.items()}
        x64 = x.double().requires_grad_(True)
        newbuf = {}
        hx, fmaps = R.encode(sdd, R._BN(sdd, train, newbuf), net, x64, dc.double() if train else None)
        if not classifier:
            return sdd, x64, {"x": hx, **{f"f{k}": f for k, f in enumerate(fmaps[1:])}}, newbuf
        pooled = hx.mean(dim=(2, 3))
        if train:
            pooled = pooled * (du.double() >= p_drop) / (1.0 - p_drop)
        return sdd, x64, {"logits": F.linear(pooled, sdd["encoder.fc.3.weight"], sdd["encoder.fc.3.bias"])}, newbuf

    for classifier in (False, True):
        plan = plan_encoder(spec, B, H, H, train, layout, pre, classifier, True, True, p_drop)
        bases = {}
        k = 2
        for base, size in (("WS", plan.ws_bytes), ("AUX", plan.aux_bytes), ("OUT", plan.out_bytes), ("X", plan.x_bytes), ("DOUT", plan.dout_bytes),
                           ("DX", plan.dx_bytes), ("NOISE", max(plan.noise_bytes, 8)), ("WPACK", getattr(plan, "wpack_bytes", 0))):
            bases[D.BASE[base]] = torch.zeros(k * ((size + 7) // 8 * 8) + 64, dtype=torch.uint8)
        bases[D.BASE["PARAMS"]] = _bytes(fp.double().clone())
        bases[D.BASE["GRADS"]] = _bytes(torch.zeros(layout.n_params, dtype=torch.float64))
        bases[D.BASE["WGS"]] = _bytes(torch.zeros(layout.n_params, dtype=torch.float64))
        bases[D.BASE["BUFS"]] = _bytes(fb.double().clone())
        bases[D.BASE["CONST"]] = _bytes(torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32))
        m = Mem(bases, True, (D.BASE["CONST"],))
        m.view(plan.inputs["x"].ref, plan.inputs["x"].shape).copy_(x.double())
        if train:
            m.view(plan.noise["drop_connect"].ref, plan.noise["drop_connect"].shape).copy_(dc.double())
            if classifier:
                m.view(plan.noise["dropout_u"].ref, plan.noise["dropout_u"].shape).copy_(du.double())
        emulate(plan.fwd.pack(), bases, True)
        sdd, x64, outs, newbuf = oracle(classifier)
        assert set(outs) == set(plan.outputs)
        tot = 0
        for j, (name, o) in enumerate(outs.items()):
            r = plan.outputs[name]
            assert tuple(r.shape) == tuple(o.shape), name
            assert rel_err(m.view(r.ref, r.shape).numpy(), o.detach().numpy()) < 1e-6, name   # (BN eps travels as f32)
            w = torch.randn(o.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(10 + j))
            tot = tot + (o * w).sum()
            d = plan.douts[name]
            m.view(d.ref, d.shape).copy_(w)
        tot.backward()
        emulate(plan.bwd.pack(), bases, True)
        gx = plan.dins["x"]
        assert rel_err(m.view(gx.ref, gx.shape).numpy(), x64.grad.numpy()) < 1e-6
        grads = fview(bases, "GRADS", True)
        scale = max(v.grad.abs().max().item() for v in sdd.values() if getattr(v, "grad", None) is not None)
        for name, (off, shape) in layout.params.items():
            full = name if nested else "encoder." + name
            ref = sdd[full].grad if full in sdd else None
            g = grads[off:off + int(np.prod(shape))].view(shape)
            if ref is None:
                assert g.abs().max() == 0, name
            else:
                assert (g - ref).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1e-4 * scale), name
        if train:
            bufs = fview(bases, "BUFS", True)
            for name, (off, shape) in layout.bufs.items():
                full = name if nested else "encoder." + name
                if full in newbuf:
                    assert rel_err(bufs[off:off + shape[0]].numpy(), newbuf[full].numpy()) < 1e-6, name


def test_bf16_mixed_plan_algebra_float64():
    """The bf16-mixed training plan in the float64 emulation (operand rounding and bf16 storage are not emulated there): its stage
    algebra - the stem as patch columns, dY tensors that live in their own (bf16) buffers between BN_BWD_APPLY and the 1x1 stages that
    read them - must reproduce the oracle's gradients to 1e-5 like the f32 plan's."""
    from s2lc_amd.plan import opdefs as D

    version, C, H, B, ncls = "b0", 6, 64, 2, 5
    model, net, sd, x, y, noise = _setup(version, C, H, B, ncls, seed=23, dcr=0.25)
    model.precision = "bf16-mixed"
    plan = model._make_plan(B, H, H, True)
    stored = [f for k, f in plan.bwd.ops if k == "BN_BWD_APPLY" and f.get("OUT_BF16")]
    readers = [f for k, f in plan.bwd.ops if (k == "CONV" and f.get("X1_BF16")) or (k == "WGRAD" and f.get("P_BF16"))]
    assert len(stored) >= 5 and len(readers) >= len(stored)       # (toy maps: the 4x4 / 2x2 stages have too few pixels for the bf16 weight-gradient kernel)
    assert all(f.get("_flags", 0) & D.FLAG_BF16 for f in readers)  # only the bf16 kernels read such a tensor
    assert all(f["DY"].dtype == "bf16" and f["DY"].ref != f["GP"].ref for f in stored)
    bases = make_bases(plan, model._flat_params, model._flat_bufs, x, noise, B * ncls * H * H, True)
    emulate(plan.fwd.pack(), bases, True)
    sd64, logits64, dlogits64, _ = oracle_grads(net, sd, x, y, noise, ncls, torch.float64)
    assert rel_err(fview(bases, "OUT", True).view(B, ncls, H, H).numpy(), logits64.numpy()) < 1e-6
    fview(bases, "DOUT", True).copy_(dlogits64.reshape(-1))
    emulate(plan.bwd.pack(), bases, True)
    grads = fview(bases, "GRADS", True)
    scale = max(v.grad.abs().max().item() for v in sd64.values() if v.requires_grad and v.grad is not None)
    for name, (off, shape) in plan.layout.params.items():
        ref = sd64[name].grad
        if ref is None or ref.abs().max().item() < 1e-9 * scale:
            continue
        g = grads[off:off + int(np.prod(shape))].view(shape)
        # (1e-4, not the 1e-5 of the f32 plan: a bf16 slot doubled holds an f32, so the stored dY tensors carry f32 rounding into the
        # cancelling BatchNorm-backward sums below them; measured worst 2e-5)
        assert rel_err(g.numpy(), ref.numpy()) < 1e-4, name
