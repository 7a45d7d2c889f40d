"""Step time of the Prithvi workloads of BASELINE.json (configs[3], configs[4]) on one GPU:
    python tools/bench_prithvi.py mae --batch 64      # MAE pre-training step, 6x224x224, mask 0.75
    python tools/bench_prithvi.py seg --batch 16 [--unfrozen]
Synthetic inputs, random-init weights; step = forward + loss + backward + fused Adam."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import s2lc_amd  # noqa: E402,F401
from s2lc_amd import _lib  # noqa: E402
from s2lc_amd.optim import FlatAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["mae", "seg"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--unfrozen", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--ops", action="store_true", help="with --profile: per-shape table of the MFMA stages")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16-mixed"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    from s2lc_amd.utils import load_untrained_prithvi
    if a.what == "mae":
        B = a.batch or 64
        model = load_untrained_prithvi(1).to(dev)
        x = torch.randn(B, 6, 1, 224, 224, device=dev)

        def step():
            opt.zero_grad()
            loss, _, _ = model(x, mask_ratio=0.75)
            loss.backward()
            opt.step()
            return loss
        flops = 59.8e9 * B
    else:
        from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
        from s2lc_amd.modules.prithvi_segmentation import PrithviSegmentationNet, PrithviSegmentationNetConfig
        from s2lc_amd.utils import _prithvi_model_args
        B = a.batch or 16
        bb = MaskedAutoencoderViT(**_prithvi_model_args(1), _decoder=False, _flat=False)
        model = PrithviSegmentationNet(PrithviSegmentationNetConfig(1, 4, 256, 1, 0.1, not a.unfrozen), backbone=bb).to(dev)
        x = torch.randn(B, 6, 1, 224, 224, device=dev)
        y = torch.randint(0, 4, (B, 224, 224), device=dev)
        from s2lc_amd.losses import CrossEntropyLoss
        lossf = CrossEntropyLoss(ignore_index=0)

        def step():
            opt.zero_grad()
            loss = lossf(model(x), y)
            loss.backward()
            opt.step()
            return loss
        flops = (875e9 if a.unfrozen else 804e9) * B
    opt = FlatAdam(model, lr=1e-4)
    model.train()
    model.precision = a.precision
    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = {"workload": f"prithvi-{a.what}" + ("-unfrozen" if a.unfrozen else ""), "batch": B, "ms_per_step": dt * 1e3, "samples_per_s": B / dt, "precision": a.precision,
           "algorithmic_tflops": flops / dt / 1e12, "loss": float(loss)}
    if a.profile:
        eng = next(iter(model._engines.values()))
        # per-kind device time of one forward + backward program
        for key, e in model._engines.items():
            if e.bwd is not None:
                eng = e
        noise = torch.rand(max(eng.plan.noise_bytes // 4, 1), device=dev)
        out = torch.empty(eng.plan.out_bytes + 256, dtype=torch.uint8, device=dev)
        n_dout = eng.plan.dout_bytes // 4 if eng.plan.dout_bytes else int(torch.Size(eng.plan.dout_shape).numel())
        dout = torch.ones(max(n_dout, 1), device=dev) * 1e-3
        bases = eng.bases(model, x, out, noise, dout=dout, grads=model._grad_buffer())
        st = torch.cuda.current_stream().cuda_stream
        kinds = {}
        for prog in (eng.fwd, eng.bwd):
            for k, (ms, cnt) in _lib.profile(prog, bases, st).items():
                kinds[k] = kinds.get(k, 0.0) + ms
        res["kernels_ms"] = {k: round(v, 3) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])}
        if a.ops:
            import collections
            from s2lc_amd.plan import opdefs as D
            names = {v: k for k, v in D.KIND.items()}
            agg = collections.OrderedDict()
            other = collections.OrderedDict()
            for tag, prog in (("fwd", eng.fwd), ("bwd", eng.bwd)):
                ms = _lib.profile_ops(prog, bases, st)
                for rec, t in zip(prog, ms):
                    kind = names[int(rec["kind"])]
                    if kind not in ("CONV", "WGRAD"):
                        e = other.setdefault((tag, kind, tuple(int(v) for v in rec["d"][:6]), tuple(int(v) for v in rec["n"][:2])), [0, 0.0])
                        e[0] += 1; e[1] += float(t)
                        continue
                    d = rec["d"]
                    g = lambda f: int(d[D.slot(kind, f)[1]])  # noqa: E731
                    if kind == "CONV":
                        key = (tag, kind, g("M"), g("C1") + g("C2"), g("KH"), g("B") * g("HO") * g("WO"), g("MODE"))
                        fl = 2.0 * g("M") * (g("C1") + g("C2")) * g("KH") * g("KW") * g("B") * g("HO") * g("WO")
                    else:
                        key = (tag, kind, g("M"), g("C"), g("KH"), g("B") * g("HO") * g("WO"), g("MODE"))
                        fl = 2.0 * g("M") * g("C") * g("KH") * g("KW") * g("B") * g("HO") * g("WO")
                    e = agg.setdefault(key, [0, 0.0, 0.0])
                    e[0] += 1; e[1] += float(t); e[2] += fl
            print("per-shape MFMA stages (sequential, no side stream):", file=sys.stderr)
            for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
                print("  %s %-5s M=%5d C=%5d k%d N=%7d mode%d  x%-3d %7.3f ms  %6.1f TF/s" % (*key, n, t, fl / t / 1e9), file=sys.stderr)
            print("other stages by (kind, d[:6], n[:2]):", file=sys.stderr)
            for key, (n, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:40]:
                print("  %s %-16s d=%s n=%s  x%-3d %7.3f ms  %7.1f us each" % (key[0], key[1], key[2], key[3], n, t, 1e3 * t / n), file=sys.stderr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
