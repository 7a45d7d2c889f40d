"""Round-2 fixtures from the *imported reference* (build container only — /root/reference never travels).

    python tests/golden/make_golden_r2.py [adam|luts|evalgrad|all]

adam      adam_steps.npz: two steps of the optimiser the reference configures
          (`torch.optim.Adam(net.parameters(), lr, weight_decay)`, /root/reference/src/train_segmentation.py:109-115, same call in
          train_mae_prithvi.py:98-104; lr / weight_decay of configs/segmentation.py:143-144 scaled up so that two steps move
          the parameters visibly) on F2's setup (the unmodified reference EfficientnetUnet b0, 224x224x6, bs 2, train mode,
          injected drop-connect noise, focal loss).  Stored per chosen tensor: the reference's gradients of both steps and
          params / exp_avg / exp_avg_sq after each step (strided subsamples, make_golden.sub) — the GPU test feeds the SAME gradients to s2k_adam_step, so the
          comparison isolates the optimiser arithmetic (bar 1e-6) from gradient noise.
luts      label_luts.npz: the reference's `get_cnes_transform(name, LABEL_MAPS[name])` (configs/cnes_labell_mappings.py:78-95)
          applied to every uint8 value, for every label-map name of configs/data_config.py:80-90.
evalgrad  unet_*_evalgrad_*.npz: gradients of the reference's modules in EVAL mode (BatchNorm on running statistics, no
          drop-connect): a well-conditioned end-to-end gradient fixture — without train-mode BatchNorm on tiny maps the
          fp32 gradient noise is ~1e-5, so the model-level bar against the reference can be 1e-3.
"""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

import ref_harness  # noqa: E402
from make_golden import build_ref_unet, checks, dc_blocks, sub  # noqa: E402
from oracle import detgen  # noqa: E402
from oracle import efficientnet_unet_ref as R  # noqa: E402

ADAM_TENSORS = ["encoder.stem.0.weight", "encoder.stem.1.weight", "encoder.blocks.1.stem.3.weight",
                "encoder.blocks.1.squeeze_excitation.3.bias", "encoder.blocks.5.final_layer.0.weight", "encoder.conv_head.1.bias",
                "up_convs.3.bias", "double_convs.2.4.bias", "double_convs.0.0.bias", "input_double_conv.0.weight",
                "out_conv1x1.weight", "out_conv1x1.bias"]
ADAM_LR, ADAM_WD = 1e-3, 0.05


def gen_adam(ns):
    tag, version, C, H, B, ncls, seed = "b0_224_train_bs2", "b0", 6, 224, 2, 4, 3      # F2 (make_golden.gen_unet)
    torch.manual_seed(0)
    model = build_ref_unet(ns, version, C, ncls, True)
    net = R.build(version, C, ncls)
    sd = detgen.fill_state(R.state_shapes(net), seed=seed)
    model.load_state_dict(sd, strict=True)
    model.train()
    x = detgen.normal(f"{tag}.x", (B, C, H, H), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, H, H), ncls, seed=seed)
    noise = detgen.uniform(f"{tag}.dc", (len(net.blocks), B), 0.0, 1.0, seed=seed)
    opt = torch.optim.Adam(model.parameters(), lr=ADAM_LR, weight_decay=ADAM_WD)       # train_segmentation.py:110-114
    fl = ns.losses.FocalLoss(alpha=torch.ones(ncls), gamma=2.0, label_smoothing=0.0, ignore_index=0)
    named = dict(model.named_parameters())
    out = {"meta": np.array([ADAM_LR, ADAM_WD, 0.9, 0.999, 1e-8]), "names": np.array(ADAM_TENSORS)}
    for step in (1, 2):
        opt.zero_grad()
        with ref_harness.injected_rand([noise[i] for i in dc_blocks(model)]):
            loss = fl(model(x), y)
        loss.backward()
        for n in ADAM_TENSORS:
            out[f"g{step}:{n}"] = sub(named[n].grad, 2048)
        opt.step()
        for n in ADAM_TENSORS:
            st = opt.state[named[n]]
            out[f"p{step}:{n}"] = sub(named[n], 2048)
            out[f"m{step}:{n}"] = sub(st["exp_avg"], 2048)
            out[f"v{step}:{n}"] = sub(st["exp_avg_sq"], 2048)
        out[f"loss{step}"] = np.array([loss.item()])
    assert named["encoder.fc.3.weight"] not in opt.state or not opt.state[named["encoder.fc.3.weight"]]   # grad None: skipped
    out["fc_untouched"] = np.array([float(torch.equal(named["encoder.fc.3.weight"].detach(), sd["encoder.fc.3.weight"]))])
    # whole-model check values after the two steps (flat order = registration order)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    out["params_ck"] = checks(flat)
    np.savez_compressed(HERE / "adam_steps.npz", **out)
    print("adam_steps written; losses", out["loss1"], out["loss2"], "fc untouched", out["fc_untouched"])


def _load_plain(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_luts():
    """configs/data_config.py itself imports sentinelhub (absent); the two mapping modules it re-exports need numpy only."""
    sys.dont_write_bytecode = True
    cnes = _load_plain("ref_cnes_maps", ref_harness.REF / "src" / "configs" / "cnes_labell_mappings.py")
    osm = _load_plain("ref_osm_maps", ref_harness.REF / "src" / "configs" / "osm_label_mapping.py")
    label_maps = {   # name -> map, as configs/data_config.py:80-90 binds them
        "osm-multiclass": osm.OSM_MULTICLASS, "osm-impervious-binary": osm.OSM_BINARY_IMPERVIOUS,
        "osm-nature-binary": osm.OSM_BINARY_NATURE, "osm-agriculture-binary": osm.OSM_BINARY_AGRICULTURE,
        "cnes-full": cnes.CNES_LABEL_MAP, "cnes-multiclass": cnes.CNES_SIMPLIFIED_MULTICLASS,
        "cnes-impervious-binary": cnes.CNES_SIMPLIFIED_BINARY_IMPERVIOUS, "cnes-nature-binary": cnes.CNES_SIMPLIFIED_BINARY_NATURE,
        "cnes-agriculture-binary": cnes.CNES_SIMPLIFIED_BINARY_AGRICULTURE,
    }
    vals = np.arange(256, dtype=np.uint8).reshape(16, 16)
    out = {}
    for name, lm in label_maps.items():
        got = cnes.get_cnes_transform(name, lm)(vals)
        out["lut:" + name] = np.asarray(got).reshape(-1).astype(np.int64)
        out["nclasses:" + name] = np.array([len(lm)])
        out["keys:" + name] = np.array(list(lm.keys()))
    np.savez_compressed(HERE / "label_luts.npz", **out)
    print("label_luts written:", {k[4:]: int(v.max()) for k, v in out.items() if k.startswith("lut:")})


def evalgrad_case(ns, tag, version, C, H, B, ncls, native, seed):
    torch.manual_seed(0)
    model = build_ref_unet(ns, version, C, ncls, native)
    net = R.build(version, C, ncls)
    sd = detgen.fill_state(R.state_shapes(net), seed=seed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = detgen.normal(f"{tag}.x", (B, C, H, H), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, H, H), ncls, seed=seed)
    logits = model(x)
    fl = ns.losses.FocalLoss(alpha=torch.ones(ncls), gamma=2.0, label_smoothing=0.0, ignore_index=0)
    loss = fl(logits, y)
    loss.backward()
    out = {"meta": np.array([C, H, B, ncls, int(native), seed]), "loss_focal": np.array([loss.item()]),
           "logits_sub": sub(logits, 4096), "logits_ck": checks(logits), "mask": logits.argmax(dim=1).to(torch.uint8).numpy()}
    tot = 0.0
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        out["grad:" + k] = sub(p.grad, 48)
        out["gradck:" + k] = checks(p.grad)
        tot += p.grad.double().pow(2).sum().item()
    out["grad_total_sq"] = np.array([tot])
    out["grad_none"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    new_sd = model.state_dict()
    out["bufs_unchanged"] = np.array([float(all(torch.equal(new_sd[k], sd[k]) for k in sd if "running" in k or "num_batches" in k))])
    np.savez_compressed(HERE / f"unet_{tag}.npz", **out)
    print(f"unet_{tag}: loss {out['loss_focal']}, |grad|^2 {tot:.6e}, {sum(1 for k in out if k.startswith('grad:'))} gradient tensors")


def gen_evalgrad(ns):
    evalgrad_case(ns, "b0_128x4_evalgrad_bs2", "b0", 4, 128, 2, 4, False, 31)
    evalgrad_case(ns, "b5_64x13_evalgrad_bs2", "b5", 13, 64, 2, 4, False, 32)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(8)
    if what in ("luts", "all"):
        gen_luts()
    if what in ("adam", "evalgrad", "all"):
        ns = ref_harness.load()
        if what in ("adam", "all"):
            gen_adam(ns)
        if what in ("evalgrad", "all"):
            gen_evalgrad(ns)
