"""Headline benchmark: EfficientNet-UNet-b5 training step on synthetic Sentinel-2 tiles.

Workload (BASELINE.json configs[1] / [2]): efficientnet-unet-b5, 13 bands, 256x256, batch 32 per
GPU, focal loss (gamma 2, ignore 0), train-mode BatchNorm, drop-connect 0.2, fp32.  One step =
forward + loss + backward (+ gradient all-reduce over RCCL when N > 1) + fused Adam step, inputs
already resident in HBM.  Prints ONE JSON line (rank 0) with the whole-job tiles/s, the roofline of
the dominant kernel family measured live with HIP events, and a bounded CPU baseline of the same
step (the CPU oracle, rank 0, N = 1 only).

  python bench.py [--gpus N] [--steps K] [--warmup W]       (N > 1: launched by torch.distributed.run)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def conv_flops(packed, D) -> dict:
    """Algorithmic FLOPs per launch family, from the stage records themselves (2*M*N*K each)."""
    names = {v: k for k, v in D.KIND.items()}
    out = {}
    for rec in packed:
        kind = names[int(rec["kind"])]
        d = rec["d"]
        if kind == "CONV":
            B, C1, C2, M, KH, KW, HO, WO = (int(d[D.slot("CONV", k)[1]]) for k in ("B", "C1", "C2", "M", "KH", "KW", "HO", "WO"))
            fl = 2.0 * M * (C1 + C2) * KH * KW * B * HO * WO
        elif kind == "WGRAD":
            B, M, C, KH, KW, HO, WO = (int(d[D.slot("WGRAD", k)[1]]) for k in ("B", "M", "C", "KH", "KW", "HO", "WO"))
            fl = 2.0 * M * C * KH * KW * B * HO * WO
        else:
            continue
        out[kind] = out.get(kind, 0.0) + fl
    return out


def conv_bytes(packed, D) -> dict:
    """Algorithmic HBM bytes per launch family: every conv-like stage reads its input(s) + weights once and writes its
    output once (SURVEY §8d convention), from the stage records."""
    names = {v: k for k, v in D.KIND.items()}
    out = {}
    for rec in packed:
        kind = names[int(rec["kind"])]
        d = rec["d"]
        if kind == "CONV":
            B, C1, C2, H, W, M, KH, KW, HO, WO, mode = (int(d[D.slot("CONV", k)[1]]) for k in ("B", "C1", "C2", "H", "W", "M", "KH", "KW", "HO", "WO", "MODE"))
            by = 4.0 * (B * (C1 + C2) * H * W + B * M * HO * WO + M * (C1 + C2) * KH * KW)
        elif kind == "WGRAD":
            B, M, C, H, W, KH, KW, HO, WO = (int(d[D.slot("WGRAD", k)[1]]) for k in ("B", "M", "C", "H", "W", "KH", "KW", "HO", "WO"))
            by = 4.0 * (B * M * HO * WO + B * C * H * W + M * C * KH * KW)
        else:
            continue
        out[kind] = out.get(kind, 0.0) + by
    return out


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC run (profiles/*_hbm_traffic.json, newest), or None."""
    files = sorted((ROOT / "profiles").glob("*_hbm_traffic.json"))
    if not files:
        return None
    try:
        return json.loads(files[-1].read_text())["kernels"][kernel]["hbm_bytes"]
    except Exception:
        return None


def cpu_baseline(version, C, H, ncls, budget_s=20.0):
    """The CPU oracle (a port of the reference's torch CPU path, validated against the reference via
    tests/golden) timed on this host: forward + backward of the same step at a small batch."""
    from oracle import detgen, losses_ref
    from oracle import efficientnet_unet_ref as R

    B = 2
    net = R.build(version, C, ncls)
    sd = detgen.fill_state(R.state_shapes(net), seed=1)
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x = torch.randn(B, C, H, H)
    y = torch.randint(0, ncls, (B, H, H))
    noise = torch.rand(len(net.blocks), B)

    def step():
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        logits = R.unet_forward(sd, net, x, training=True, dc_noise=noise)
        losses_ref.focal(logits, y, torch.ones(ncls), 2.0, 0.0, ignore_index=0).backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} fwd+bwd steps of efficientnet-unet-{version} {C}x{H}x{H} at batch {B} (oracle/, torch CPU fp32 eager)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--version", default="b5")
    ap.add_argument("--bands", type=int, default=13)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # one rank per GPU; a rehearsal of the N-rank code path on a single-GPU box maps every rank onto device 0
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    import s2lc_amd  # noqa: F401
    from s2lc_amd import _lib
    from s2lc_amd.losses import FocalLoss
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
    from s2lc_amd.optim import FlatAdam
    from s2lc_amd.plan import opdefs as D

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("S2K_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ncls = 4
    torch.manual_seed(42)  # identical initial weights on every rank (configs/segmentation.py:103 seed)
    model = EfficientnetUnet(EfficientNetConfig(args.version, args.bands, ncls, class_distribution=[0.25] * ncls))
    model.to(dev).train()
    opt = FlatAdam(model, lr=1.5e-6, weight_decay=0.05)  # BASE_CONFIG lr / weight_decay
    loss_fn = FocalLoss(torch.ones(ncls), 2.0, 0.0, ignore_index=0)
    ddp = None
    if world > 1:
        from s2lc_amd.ddp import FlatGradReducer

        ddp = FlatGradReducer(model, dist)
        ddp.broadcast_parameters(0)

    g = torch.Generator(device=dev).manual_seed(42 + rank)
    B, C, H = args.batch, args.bands, args.size
    x = torch.randn(B, C, H, H, device=dev, generator=g)
    y = torch.randint(1, ncls, (B, H, H), device=dev, generator=g)
    y[torch.rand(B, H, H, device=dev, generator=g) < 0.05] = 0  # ~5 % ignored pixels

    def step():
        opt.zero_grad()
        logits = model(x)
        loss = loss_fn(logits, y)
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())

    roofline, kernels, cpu = None, None, None
    if rank == 0:
        if not args.no_profile:
            # live per-kernel-family device time (HIP events on the launch stream) of one fwd+bwd
            eng = next(iter(model._engines.values()))
            st = torch.cuda.current_stream().cuda_stream
            out = torch.empty(eng.plan.logits_shape, device=dev)
            noise = torch.rand(eng.n_noise_rows, B, device=dev)
            grads = model._grad_buffer()
            dout = torch.randn(eng.plan.logits_shape, device=dev) * 1e-6
            pf = _lib.profile(eng.fwd, eng.bases(model, x, out, noise=noise), st)
            pb = _lib.profile(eng.bwd, eng.bases(model, x, None, dout=dout, noise=noise, grads=grads), st)
            fl_f, fl_b = conv_flops(eng.fwd, D), conv_flops(eng.bwd, D)
            kernels = {}
            for nm in sorted(set(pf) | set(pb)):
                ms = pf.get(nm, (0, 0))[0] + pb.get(nm, (0, 0))[0]
                cnt = pf.get(nm, (0, 0))[1] + pb.get(nm, (0, 0))[1]
                kernels[nm] = {"ms": round(ms, 3), "launches": cnt}
            conv_ms = kernels["CONV"]["ms"]
            conv_fl = fl_f.get("CONV", 0.0) + fl_b.get("CONV", 0.0)
            wg_ms = kernels.get("WGRAD", {"ms": 0.0})["ms"]
            wg_fl = fl_b.get("WGRAD", 0.0)
            dom, dms, dfl = ("conv_igemm_kernel", conv_ms, conv_fl) if conv_ms >= wg_ms else ("wgrad_kernel", wg_ms, wg_fl)
            n_l = kernels["CONV" if dom == "conv_igemm_kernel" else "WGRAD"]["launches"]
            ach = dfl / (dms * 1e-3) / 1e12
            by_f, by_b = conv_bytes(eng.fwd, D), conv_bytes(eng.bwd, D)
            dkind = "CONV" if dom == "conv_igemm_kernel" else "WGRAD"
            alg_bytes = (by_f.get(dkind, 0.0) + by_b.get(dkind, 0.0)) / n_l
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        # HBM bytes per launch from the PMC counters (committed run, see profiles/); algorithmic bytes beside it
                        "traffic": measured_traffic(dom), "algorithmic_bytes_per_launch": round(alg_bytes),
                        "launches": n_l, "avg_launch_ms": round(dms / n_l, 4),
                        "flops_per_step": dfl, "all_mfma_tflops": round((conv_fl + wg_fl) / ((conv_ms + wg_ms) * 1e-3) / 1e12, 2)}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.version, C, H, ncls)
        tiles = world * B * args.steps
        line = {
            "metric": "Sentinel-2 256x256x13 tiles/sec fwd+bwd", "value": round(tiles / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"efficientnet-unet-{args.version} {C}x{H}x{H} bs{B}/GPU focal(g=2) train step "
                                   f"(fwd+loss+bwd{'+allreduce' if world > 1 else ''}+adam)",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "loss": round(loss_val, 6),
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()          # rank 0 profiles after the timed region: nobody tears the group down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
