"""Per-stage device time of one training step (HIP events), with achieved TFLOP/s (MFMA stages) or GB/s."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import s2lc_amd  # noqa: E402,F401
from s2lc_amd import _lib  # noqa: E402
from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet  # noqa: E402
from s2lc_amd.plan import opdefs as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--version", default="b5")
    ap.add_argument("--bands", type=int, default=13)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--kinds", default="CONV,WGRAD")
    ap.add_argument("--top", type=int, default=400)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--order", action="store_true", help="program order instead of slowest first")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = EfficientnetUnet(EfficientNetConfig(a.version, a.bands, 4, class_distribution=[.25] * 4)).to(dev).train()
    model.precision = a.precision
    x = torch.randn(a.batch, a.bands, a.size, a.size, device=dev)
    model(x)  # builds the engine, warms up
    eng = next(iter(model._engines.values()))
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(eng.plan.logits_shape, device=dev)
    noise = torch.rand(eng.n_noise_rows, a.batch, device=dev)
    grads = model._grad_buffer()
    dout = torch.randn(eng.plan.logits_shape, device=dev) * 1e-6
    names = {v: k for k, v in D.KIND.items()}
    kinds = set(a.kinds.split(","))
    for tag, packed, bases in (("fwd", eng.fwd, eng.bases(model, x, out, noise=noise)),
                               ("bwd", eng.bwd, eng.bases(model, x, None, dout=dout, noise=noise, grads=grads))):
        _lib.profile_ops(packed, bases, st)
        ms, var = _lib.profile_variants(packed, bases, st)
        rows = []
        for i, rec in enumerate(packed):
            kind = names[int(rec["kind"])]
            if kind not in kinds and "ALL" not in kinds:
                continue
            d = rec["d"]
            if kind == "CONV":
                g = {k: int(d[D.slot("CONV", k)[1]]) for k in ("B", "C1", "C2", "M", "KH", "H", "W", "HO", "WO", "MODE", "STRIDE", "PRO1", "PRO2")}
                fl = 2.0 * g["M"] * (g["C1"] + g["C2"]) * g["KH"] ** 2 * g["B"] * g["HO"] * g["WO"]
                gate = int(rec["t"][D.slot("CONV", "GATE1")[1]]) >= 0
                desc = (f"M={g['M']:5d} C={g['C1']}+{g['C2']} k{g['KH']} s{g['STRIDE']} {g['HO']}x{g['WO']} mode{g['MODE']} pro{g['PRO1']}{g['PRO2']}"
                        f"{' gate' if gate else ''} {('generic', 'PC', 'BF16', 'DMA', 'Q4')[var[i]]}")
                by = 4.0 * (g["B"] * (g["C1"] + g["C2"]) * g["H"] * g["W"] + g["B"] * g["M"] * g["HO"] * g["WO"])
            elif kind == "WGRAD":
                g = {k: int(d[D.slot("WGRAD", k)[1]]) for k in ("B", "M", "C", "KH", "HO", "WO", "MODE", "PROP", "PROQ")}
                fl = 2.0 * g["M"] * g["C"] * g["KH"] ** 2 * g["B"] * g["HO"] * g["WO"]
                gate = int(rec["t"][D.slot("WGRAD", "GATEP")[1]]) >= 0 or int(rec["t"][D.slot("WGRAD", "GATEQ")[1]]) >= 0
                desc = (f"M={g['M']:5d} C={g['C']} k{g['KH']} {g['HO']}x{g['WO']} mode{g['MODE']} pro{g['PROP']}{g['PROQ']}"
                        f"{' gate' if gate else ''} {('generic', 'PC', 'BF16', 'DMA', 'Q4')[var[i]]}")
                by = 4.0 * g["B"] * g["HO"] * g["WO"] * (g["M"] + g["C"])
            else:
                fl, desc, by = 0.0, " ".join(str(int(v)) for v in d[:11]), 0.0
            rows.append((float(ms[i]), tag, i, kind, desc, fl, by))
        rows.sort(key=(lambda r: r[2]) if a.order else None, reverse=not a.order)
        tot = sum(r[0] for r in rows)
        print(f"== {tag}: {tot:.2f} ms in {len(rows)} stages of kinds {sorted(kinds)}")
        for t, tg, i, kind, desc, fl, by in rows[: a.top]:
            tf = fl / (t * 1e-3) / 1e12 if t > 0 and fl else 0.0
            gb = by / (t * 1e-3) / 1e9 if t > 0 else 0.0
            print(f"{t:8.3f} ms  {tg} #{i:4d} {kind:14s} {desc:78s} {tf:7.1f} TF/s {gb:7.0f} GB/s")


if __name__ == "__main__":
    main()
