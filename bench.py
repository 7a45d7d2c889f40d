"""Headline benchmark: EfficientNet-UNet-b5 training step on synthetic Sentinel-2 tiles.

Workload (BASELINE.json configs[1] / [2]): efficientnet-unet-b5, 13 bands, 256x256, batch 32 per
GPU, focal loss (gamma 2, ignore 0), train-mode BatchNorm, drop-connect 0.2, fp32.  One step =
forward + loss + backward (+ gradient all-reduce over RCCL when N > 1) + fused Adam step, inputs
already resident in HBM.  Prints ONE JSON line (rank 0) with the whole-job tiles/s, the roofline of
the dominant kernel measured live with HIP events (spec peak and the peak measured on this GPU in the
same run), the Adam time, a bounded CPU baseline of the same step (the CPU oracle, rank 0, N = 1
only), and - N = 1 only, skipped with --no-prithvi - the Prithvi workloads of BASELINE.json
configs[3] / [4] as extra keys (`prithvi_mae`, `prithvi_seg_frozen`, `prithvi_seg_unfrozen`); the
headline `value` is always configs[1].  N > 1 adds `allreduce` (what the gradient all-reduce costs) and
`prithvi_mae` = the data-parallel Prithvi-100M MAE step (configs[4]) on the same ranks, under a watchdog.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or called plainly as `python bench.py --gpus N ...`: the parent process then
starts N rank processes itself BEFORE it touches the GPU (it never initialises HIP), relays rank 0's single JSON line and
exits with the worst child exit code - the way Lightning starts the reference's ranks from `pl.Trainer(devices=N)`
(/root/reference/src/train_segmentation.py:273-280).
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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL over dmabuf IPC (the host driver supports no other mode)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (the ~5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
GUIDE_COPY_GBS = 6290.0     # MI355X_MICROARCH.md: 6.29 TB/s measured with a float4 copy (79 % of the 8 TB/s spec)
BUCKET_MB = None           # --bucket-mb (None: ddp.DEFAULT_BUCKET_MB)
MAE_DP_LIMIT_S = float(os.environ.get("S2K_MAE_DP_LIMIT_S", "240"))   # watchdog of the N > 1 Prithvi leg (an extra key must never cost the headline line)
# SURVEY §8d, whole-step denominators (fp32): the layer-wise roofline T = sum over layers of max(FLOP / peak, bytes / BW) of
# efficientnet-unet-b5 13x256x256 is 0.464 ms per tile (pure MFMA 0.418, pure HBM 0.140) = 2,154 tiles/s per GPU; the Prithvi
# steps are compute-bound: forward + backward FLOP per sample / the f32 MFMA peak.
UNET_B5_LAYERWISE_TILES_PER_S = 2154.0
PRITHVI_GFLOP_PER_SAMPLE = {"mae": 59.8, "seg_frozen": 804.0, "seg_unfrozen": 875.0}


KERNEL_OF = {("CONV", 0): "conv_igemm_kernel", ("CONV", 1): "conv_pc_kernel", ("WGRAD", 0): "wgrad_kernel", ("WGRAD", 1): "wgrad_pc_kernel",
             ("CONV", 2): "conv_bf16_kernel", ("WGRAD", 2): "wgrad_bf16_kernel",      # variant 2: bf16 MFMA operands (bf16-mixed plans only)
             ("CONV", 3): "conv_dma_kernel",                                            # variant 3: the LDS-DMA ring kernel (csrc/conv_dma.hip)
             ("CONV", 4): "conv_q4_kernel", ("WGRAD", 4): "wgrad_q4_kernel"}            # variant 4: the quad-read 1x1 kernels (csrc/conv_q4.hip, wgrad_q4.hip)


def stage_work(rec, D):
    """(kind, algorithmic FLOPs, algorithmic HBM bytes) of one CONV / WGRAD stage record, else (kind, 0, 0): 2*M*N*K flops;
    bytes = every operand read once + the result written once (SURVEY §8d convention)."""
    kind = D.NAME_OF[int(rec["kind"])]
    d = rec["d"]
    if kind == "CONV":
        B, C1, C2, H, W, M, KH, KW, HO, WO = (int(d[D.slot("CONV", k)[1]]) for k in ("B", "C1", "C2", "H", "W", "M", "KH", "KW", "HO", "WO"))
        return kind, 2.0 * M * (C1 + C2) * KH * KW * B * HO * WO, 4.0 * (B * (C1 + C2) * H * W + B * M * HO * WO + M * (C1 + C2) * KH * KW)
    if kind == "WGRAD":
        B, M, C, H, W, KH, KW, HO, WO = (int(d[D.slot("WGRAD", k)[1]]) for k in ("B", "M", "C", "H", "W", "KH", "KW", "HO", "WO"))
        return kind, 2.0 * M * C * KH * KW * B * HO * WO, 4.0 * (B * M * HO * WO + B * C * H * W + M * C * KH * KW)
    return kind, 0.0, 0.0


def profile_programs(_lib, D, programs, bases_list, stream):
    """Device time of every stage of `programs` (HIP events on the launch stream, stages run back to back on that one stream).
    Returns (per stage-kind {ms, launches}, per MFMA kernel name {ms, launches, flops, bytes})."""
    kinds, kernels = {}, {}
    for prog, bases in zip(programs, bases_list):
        ms, var = _lib.profile_variants(prog, bases, stream)
        for rec, t, v in zip(prog, ms, var):
            kind, fl, by = stage_work(rec, D)
            e = kinds.setdefault(kind, {"ms": 0.0, "launches": 0})
            e["ms"] += float(t); e["launches"] += 1
            if fl:
                k = kernels.setdefault(KERNEL_OF[(kind, int(v))], {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
                k["ms"] += float(t); k["launches"] += 1; k["flops"] += fl; k["bytes"] += by
    return kinds, kernels


def csrc_fingerprint() -> str:
    """Hash of the kernel sources: a committed PMC traffic figure is only quoted for the sources it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / "sentinel2-landcover-classification_amd" / "csrc").iterdir()):
        if f.suffix in (".hip", ".h"):                 # sources only (the objects built beside them - *.hip.o - differ from box to box)
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def make_roofline(kernels: dict, peaks: dict | None, traffic_workload: bool = True) -> dict:
    """Roofline object of the MFMA kernel with the most device time.  `traffic_workload`: the committed PMC run measured THIS
    workload (the U-Net headline step); other workloads launch the same kernel names on other shapes and get no figure."""
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    k = kernels[dom]
    ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
    tot_fl = sum(v["flops"] for v in kernels.values())
    tot_ms = sum(v["ms"] for v in kernels.values())
    traffic, src = measured_traffic(dom) if traffic_workload else (None, "none: the committed PMC run covers the U-Net headline step only")
    r = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
         # HBM bytes per launch from the PMC counters of a COMMITTED rocprofv3 run (not this run): null when that file was
         # measured on other kernel sources than the ones built here
         "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"]),
         "launches": k["launches"], "avg_launch_ms": round(k["ms"] / k["launches"], 4), "flops_per_step": k["flops"],
         "all_mfma_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
         "mfma_kernels": {n: {"ms": round(v["ms"], 3), "launches": v["launches"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                              "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]),
                              "traffic": measured_traffic(n, quiet=True)[0] if traffic_workload else None}
                          for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}}
    if peaks:
        r["peak_measured"] = round(peaks["mfma_f32_tflops"], 1)
        r["frac_of_measured"] = round(ach / peaks["mfma_f32_tflops"], 4)
    return r


def measured_traffic(kernel: str, quiet: bool = False):
    """(HBM bytes per launch of `kernel`, provenance) from the newest committed PMC run (profiles/*_hbm_traffic.json, written by
    tools/pmc_traffic.sh + tools/traffic_json.py).  The figure is NOT measured in this run (PMC counters need rocprofv3 around
    the process); it is dropped - loudly - when the file was measured on kernel sources other than the ones built here."""
    files = sorted((ROOT / "profiles").glob("*_hbm_traffic.json"))
    if not files:
        return None, "none: no profiles/*_hbm_traffic.json"
    f = files[-1]
    try:
        doc = json.loads(f.read_text())
        have, want = doc.get("csrc_fingerprint"), csrc_fingerprint()
        if have != want:
            if not quiet:
                print(f"bench.py: {f.name} was measured on csrc {have}, this tree is {want}: roofline.traffic omitted "
                      f"(re-run tools/pmc_traffic.sh)", file=sys.stderr)
            return None, f"stale: profiles/{f.name} (csrc {have}, built {want})"
        return doc["kernels"][kernel]["hbm_bytes"], f"committed PMC run profiles/{f.name} (same csrc {want}); not measured in this run"
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable: profiles/{f.name}: {e}"


def cpu_model() -> dict:
    """CPU model string, sockets and cores of this host."""
    model, phys, cores = None, set(), set()
    try:
        pid = None
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip(); phys.add(pid)
            elif line.startswith("core id"):
                cores.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    quota = None
    try:        # cgroup v2 CPU bandwidth of this container: "max 100000" or "<quota> <period>"
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    return {"model": model, "sockets": len(phys) or None, "physical_cores": len(cores) or None, "logical_cpus": os.cpu_count(),
            "usable_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota}


def cpu_baseline(version, C, H, ncls, steps=5):
    """The CPU oracle (a port of the reference's torch CPU path, validated against the reference via tests/golden) timed on
    this host: forward + backward of the same step.  torch's intra-op thread count is swept over {16, 32, 64, all usable}
    on half-size tiles at batch 2 (one warm-up + one timed step each: the sweep only picks the thread count), then `steps`
    timed steps at the best count on the full-size tiles at batch 2 and at batch 4; the better of the two is reported.  The sweep runs in ascending order and
    stops as soon as the rate has fallen to half of the best seen (thread counts beyond the container's CPU share)."""
    from oracle import detgen, losses_ref
    from oracle import efficientnet_unet_ref as R

    net = R.build(version, C, ncls)
    sd = detgen.fill_state(R.state_shapes(net), seed=1)
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)

    def make_step(B, size):
        x = torch.randn(B, C, size, size)
        y = torch.randint(0, ncls, (B, size, size))
        noise = torch.rand(len(net.blocks), B)

        def step():
            for v in sd.values():
                if v.requires_grad:
                    v.grad = None
            logits = R.unet_forward(sd, net, x, training=True, dc_noise=noise)
            losses_ref.focal(logits, y, torch.ones(ncls), 2.0, 0.0, ignore_index=0).backward()
        return step

    def say(msg):       # progress on stderr: the CPU leg is the long, silent part of a default run
        print(f"bench.py cpu_baseline: {msg}", file=sys.stderr, flush=True)

    usable = len(os.sched_getaffinity(0))
    before = torch.get_num_threads()
    sweep = {}
    hs = max(H // 2, 32)
    probe = make_step(2, hs)
    for t in sorted({min(t, usable) for t in (16, 32, 64, usable)}):
        torch.set_num_threads(t)
        probe()
        t0 = time.perf_counter()
        probe()
        sweep[t] = round(2 / (time.perf_counter() - t0), 3)
        say(f"{t} threads: {sweep[t]} tiles/s on {hs}x{hs} tiles")
        if sweep[t] < 0.5 * max(sweep.values()):
            # past the container's CPU share more threads only fight each other (a 128-thread step on a 16-CPU share did not
            # finish in 7 minutes): stop the sweep once the rate has halved
            say("rate halved: larger thread counts skipped")
            break
    best_t = max(sweep, key=sweep.get)
    torch.set_num_threads(best_t)
    by_batch = {}
    for B in (2, 4):
        st = make_step(B, H)
        st()
        t0 = time.perf_counter()
        for i in range(steps):
            st()
            say(f"batch {B} step {i + 1}/{steps} at {best_t} threads: {B * (i + 1) / (time.perf_counter() - t0):.3f} tiles/s")
        by_batch[B] = round(B * steps / (time.perf_counter() - t0), 3)
    torch.set_num_threads(before)
    best_b = max(by_batch, key=by_batch.get)
    return {"value": by_batch[best_b], "unit": "tiles/s", "cores": best_t, "kind": "port", "cpu": cpu_model(),
            "thread_sweep_tiles_per_s": {str(k): v for k, v in sweep.items()}, "thread_sweep_tile": f"{C}x{hs}x{hs} bs2",
            "by_batch_tiles_per_s": {str(k): v for k, v in by_batch.items()},
            "sample": f"{steps} timed fwd+bwd steps (after 1 warm-up) of efficientnet-unet-{version} {C}x{H}x{H} at batch {best_b} with "
                      f"{best_t} torch threads - the best of batch (2, 4); thread count picked by a sweep over {sorted(sweep)} on "
                      f"{hs}x{hs} tiles (oracle/, torch CPU fp32 eager)"}


def time_adam(opt, dev, iters=10) -> float:
    """Milliseconds of one fused Adam step over the flat parameter buffer (HIP events on the current stream).  Called on rank 0 alone:
    in the sharded mode the kernels over the owned slices are timed without the parameter all-gather (a collective)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    opt.step(gather=False)
    e0.record()
    for _ in range(iters):
        opt.step(gather=False)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def prithvi_workload(what: str, dev, peaks, steps=8, warmup=3) -> dict:
    """One Prithvi workload of BASELINE.json (configs[3]: MAE pre-training bs 64 mask 0.75; configs[4]: segmentation fine-tuning
    bs 16, frozen / unfrozen backbone), 6x1x224x224 synthetic inputs, random-init Prithvi-100M: forward + loss + backward + Adam."""
    from s2lc_amd import _lib
    from s2lc_amd.optim import FlatAdam
    from s2lc_amd.plan import opdefs as D
    from s2lc_amd.utils import _prithvi_model_args, load_untrained_prithvi

    torch.manual_seed(42)
    if what == "mae":
        B = 64
        model = load_untrained_prithvi(1).to(dev)
        x = torch.randn(B, 6, 1, 224, 224, device=dev)

        def fwd_loss():
            return model(x, mask_ratio=0.75)[0]
        name = "prithvi-100M MAE pre-training step 6x1x224x224 bs64 mask 0.75 (fwd+loss+bwd+adam)"
    else:
        from s2lc_amd.losses import CrossEntropyLoss
        from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
        from s2lc_amd.modules.prithvi_segmentation import PrithviSegmentationNet, PrithviSegmentationNetConfig

        B = 16
        frozen = what == "seg_frozen"
        bb = MaskedAutoencoderViT(**_prithvi_model_args(1), _decoder=False, _flat=False)
        model = PrithviSegmentationNet(PrithviSegmentationNetConfig(1, 4, 256, 1, 0.1, frozen), backbone=bb).to(dev)
        x = torch.randn(B, 6, 1, 224, 224, device=dev)
        y = torch.randint(0, 4, (B, 224, 224), device=dev)
        lossf = CrossEntropyLoss(ignore_index=0)

        def fwd_loss():
            return lossf(model(x), y)
        name = f"prithvi-100M segmentation fine-tuning step 6x1x224x224 bs16 CE(ignore 0), {'frozen' if frozen else 'unfrozen'} backbone (fwd+loss+bwd+adam)"
    opt = FlatAdam(model, lr=1e-4)
    model.train()

    def step():
        opt.zero_grad()
        loss = fwd_loss()
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    eng = [e for e in model._engines.values() if getattr(e, "bwd", None) is not None][-1]
    noise = torch.rand(max(eng.plan.noise_bytes // 4, 1), device=dev)
    out = torch.empty(eng.plan.out_bytes + 256, dtype=torch.uint8, device=dev)
    n_dout = eng.plan.dout_bytes // 4 if eng.plan.dout_bytes else int(torch.Size(eng.plan.dout_shape).numel())
    dout = torch.ones(max(n_dout, 1), device=dev) * 1e-3
    bases = eng.bases(model, x, out, noise, dout=dout, grads=model._grad_buffer())
    _, kernels = profile_programs(_lib, D, (eng.fwd, eng.bwd), (bases, bases), torch.cuda.current_stream().cuda_stream)
    alg_tf = PRITHVI_GFLOP_PER_SAMPLE[what] * 1e9 * B / dt / 1e12
    # the same workload in the separately reported bf16-mixed mode (Linears / neck / head convs and their weight gradients on bf16
    # MFMA operands; attention, LayerNorm, GELU, BatchNorm statistics, loss, master weights, Adam in f32)
    b16 = None
    try:
        roof32 = make_roofline(kernels, peaks, traffic_workload=False)
        f32_loss = float(loss.detach())
        model.precision = "bf16-mixed"
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss16 = step()
        torch.cuda.synchronize(dev)
        dt16 = (time.perf_counter() - t0) / steps
        b16 = {"value": round(B / dt16, 2), "unit": "samples/s", "ms_per_step": round(dt16 * 1e3, 3), "dtype": "bf16-mixed",
               "speedup_vs_f32": round(dt / dt16, 3), "loss": round(float(loss16.detach()), 6),
               "step_algorithmic_tflops": round(PRITHVI_GFLOP_PER_SAMPLE[what] * 1e9 * B / dt16 / 1e12, 1),
               "step_frac_of_bf16_mfma_peak": round(PRITHVI_GFLOP_PER_SAMPLE[what] * 1e9 * B / dt16 / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
               "note": "weights have moved by the f32 steps before it; parity of the mode: tests/test_bf16_mixed_gpu.py"}
        model.precision = "f32"
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: prithvi_{what} bf16-mixed leg failed: {e!r}", file=sys.stderr, flush=True)
        b16 = {"error": repr(e)[:300]}
        roof32 = make_roofline(kernels, peaks, traffic_workload=False)
        f32_loss = float(loss.detach())
    return {"bf16_mixed": b16, "workload": name, "value": round(B / dt, 2), "unit": "samples/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup,
            "batch": B, "dtype": "f32", "step_algorithmic_tflops": round(alg_tf, 1), "step_frac_of_mfma_peak": round(alg_tf / PEAK_F32_MFMA_TFLOPS, 4), "adam_ms": round(time_adam(opt, dev), 4), "loss": round(f32_loss, 6), "roofline": roof32}


def launch_ranks(n: int, argv: list[str], script: str | None = None) -> int:
    """`python bench.py --gpus N` with no rank environment: start N fresh rank processes (one per GPU) of this same script and
    relay their output.  This parent must not have initialised the GPU (it only imported torch) and never does: replacing or
    forking a process that holds a HIP context is what takes a node down.  Returns the worst child exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # rank 0's stdout is filtered (below): its one JSON line IS this program's output; the other ranks print nothing there
        procs.append(subprocess.Popen([sys.executable, script or str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # rank 0's stdout is relayed by a reader THREAD while the main loop polls every rank: a rank k > 0 that dies early (out of
    # memory, import error, RCCL init failure) leaves rank 0 inside init_process_group or a collective, its stdout open - a parent
    # that first reads rank 0 to EOF would sit there until rank 0's own distributed timeout.  First non-zero exit: the
    # other ranks are ended (exact PIDs, ours) and that code is returned; S2K_LAUNCH_DEADLINE_S (default 1500) bounds the whole run.
    import threading

    def relay(pipe):
        # libraries chat on stdout too (gloo: "[Gloo] Rank 0 is connected to ..."): only the result line goes to this program's
        # stdout, everything else rank 0 printed there is passed on through stderr
        for ln in pipe:
            is_result = False
            if ln.lstrip().startswith("{"):
                try:
                    is_result = isinstance(json.loads(ln), dict)
                except ValueError:
                    pass
            (sys.stdout if is_result else sys.stderr).write(ln)
            (sys.stdout if is_result else sys.stderr).flush()

    reader = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("S2K_LAUNCH_DEADLINE_S", "1500"))
    worst = 0

    def end_all():
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 10.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()

    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"bench.py: rank {r} exited with code {c}: ending the other ranks", file=sys.stderr, flush=True)
                worst = c
                end_all()
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                print("bench.py: ranks still running at the launch deadline (S2K_LAUNCH_DEADLINE_S): ending them", file=sys.stderr, flush=True)
                worst = 124
                end_all()
                break
            time.sleep(0.05)
        reader.join(timeout=5.0)
    except BaseException:
        for q in procs:
            if q.poll() is None:
                q.kill()
        raise
    return worst if worst >= 0 else 1


def time_steps(step, n: int, dist, dev) -> float:
    """Seconds for n steps, bracketed by barrier + synchronize on both sides, MAX over ranks."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64) if dist.get_backend() == "nccl" else torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def allreduce_report(model, ddp, dist, dev, world: int, step_ms: float, noop_ms: float, calls: list) -> dict:
    """What the gradient all-reduce costs (SURVEY.md §8d evidence for configs[2]): bucket count and bytes per step, the time
    it adds to the step (`ms_exposed` = step with the reducer - the same segmented step with the hook replaced by a no-op,
    i.e. what the overlap did NOT hide), and the bus bandwidth of the same buckets reduced back to back on an otherwise idle
    GPU (2 (N-1)/N x bytes / time, the nccl-tests convention)."""
    grads = model._grad_buffer()
    buckets = [(lo, hi) for lo, hi in calls if hi > lo]
    nbytes = 4 * sum(hi - lo for lo, hi in buckets)
    iters = 5

    sharded = getattr(ddp, "mode", "allreduce") == "sharded"

    def alone():
        for lo, hi in buckets:
            ddp.on_segment(lo, hi, grads)
        ddp.finish()
        if sharded:     # the second half of the collective: the parameters of every bucket gathered from their owners
            ddp.all_gather_slices(model._flat_params)
    alone()
    dt = time_steps(alone, iters, dist, dev) / iters
    bus = 2.0 * (world - 1) / world * nbytes / dt / 1e9
    return {"buckets": len(buckets), "bytes": nbytes, "bucket_bytes": [4 * (hi - lo) for lo, hi in buckets],
            "bucket_mb_planned": round(float(getattr(ddp, "bucket_mb", 0.0)), 3),      # ddp.DEFAULT_BUCKET_MB / --bucket-mb: the planner's bucket size
            "ms_exposed": round(step_ms - noop_ms, 3), "ms_step_with_reducer": round(step_ms, 3), "ms_step_noop_hook": round(noop_ms, 3),
            "ms_alone": round(dt * 1e3, 3), "bus_gbps": round(bus, 1), "overlap_fraction": round(max(0.0, 1.0 - max(step_ms - noop_ms, 0.0) / (dt * 1e3)), 3),
            "backend": dist.get_backend(), "mode": "sharded" if sharded else "allreduce",
            "op": ("reduce_scatter(SUM) per contiguous suffix bucket of the flat gradient buffer on a side stream, Adam on the owned slices, "
                   "all_gather of the updated parameters per bucket (--sharded-adam)") if sharded else
                  "all_reduce(SUM) per contiguous suffix bucket of the flat gradient buffer, side stream"}


def prithvi_mae_dp_leg(dist, dev, rank: int, world: int, batch: int = 64, steps: int = 5, warmup: int = 2) -> dict:
    """BASELINE.json configs[4]'s data-parallel leg (Prithvi-100M MAE pre-training, bs 64 per GPU, mask 0.75) on the N ranks of this
    run: the same `FlatGradReducer` (segmented backward, suffix buckets of the flat gradient buffer all-reduced on a side stream),
    timed under the headline's protocol (barrier + synchronize on both sides, MAX over ranks).  Every rank runs it; rank 0 reports."""
    from s2lc_amd.ddp import FlatGradReducer
    from s2lc_amd.optim import FlatAdam
    from s2lc_amd.utils import load_untrained_prithvi

    torch.manual_seed(42)
    model = load_untrained_prithvi(1).to(dev)
    model.train()
    opt = FlatAdam(model, lr=1e-4)
    ddp = FlatGradReducer(model, dist)
    ddp.broadcast_parameters(0)
    x = torch.randn(batch, 6, 1, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(1042 + rank))

    def step():
        opt.zero_grad()
        loss = model(x, mask_ratio=0.75)[0]
        loss.backward()
        ddp.finish()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    dt = time_steps(step, steps, dist, dev)
    calls = []
    model._bwd_segment_hook = lambda lo, hi, grads: calls.append((lo, hi))     # the same segmented backward, no collective
    step()
    calls.clear()
    loss = step()
    buckets = list(calls)
    noop_dt = time_steps(step, steps, dist, dev)
    model._bwd_segment_hook = ddp.on_segment
    try:
        ar = allreduce_report(model, ddp, dist, dev, world, dt / steps * 1e3, noop_dt / steps * 1e3, buckets)
    except Exception as e:  # noqa: BLE001
        ar = {"error": repr(e)[:300]}
    return {"workload": f"prithvi-100M MAE pre-training step 6x1x224x224 bs{batch}/GPU mask 0.75 (fwd+loss+bwd+allreduce+adam)",
            "value": round(world * batch * steps / dt, 2), "unit": "samples/s", "n_gpus": world, "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "warmup": warmup, "global_batch": world * batch, "parallelism": f"dp{world}", "scaling": "weak", "dtype": "f32",
            "n1_same_plan_samples_per_s": round(batch * steps / noop_dt, 2), "allreduce": ar, "loss": round(float(loss.item()), 6)}


def bf16_mixed_leg(model, cfg, x, y, loss_fn, args, dev, f32_tiles_per_s, with_oracle, _lib, D) -> dict:
    """The bf16-MIXED mode, reported separately (never instead of the f32 headline): the arithmetic class the reference itself
    trains with (`precision="bf16"`, /root/reference/src/configs/segmentation.py:146,153).  Same network, same weights as the f32
    model at this point, same batch: dense convs / weight gradients round their MFMA operands to bf16 (csrc/conv_bf16.hip,
    wgrad_bf16.hip), everything else - accumulation, BatchNorm statistics, loss, master weights, Adam - stays f32.
    Returns throughput under the headline's timing protocol, the roofline of its dominant kernel and a parity table."""
    from s2lc_amd.losses import class_mask
    from s2lc_amd.modules.efficientnet_unet import EfficientnetUnet
    from s2lc_amd.optim import FlatAdam

    B = x.shape[0]
    m16 = EfficientnetUnet(cfg)
    m16.load_state_dict(model.state_dict())
    m16.to(dev).train()
    m16.precision = "bf16-mixed"
    # ---- parity: one training step of both models from identical weights, batch and drop-connect noise -----------------------
    eng = next(iter(model._engines.values()))
    noise = torch.rand(eng.n_noise_rows, B, device=dev)
    res = {}
    for tag, m in (("f32", model), ("b16", m16)):
        bufs = m._flat_bufs.clone()
        nbt = m._flat_nbt.clone()
        m.drop_connect_noise = noise
        for p in m.parameters():
            p.grad = None
        logits = m(x)
        loss = loss_fn(logits, y)
        loss.backward()
        torch.cuda.synchronize(dev)
        res[tag] = (logits.detach().clone(), float(loss), m._grad_buffer().detach().clone())
        m._flat_bufs.copy_(bufs)          # the running statistics of the comparison step are not kept
        m._flat_nbt.copy_(nbt)
        m.drop_connect_noise = None
        for p in m.parameters():
            p.grad = None
    (l32, s32, g32), (l16, s16, g16) = res["f32"], res["b16"]
    d = (l16 - l32).double()
    parity = {"reference": "this library's f32 path on the same GPU, same weights / batch / drop-connect noise (train-mode BatchNorm)",
              "logits_max_rel_err": float(d.abs().max() / l32.abs().max()),
              "logits_rms_rel_err": float(d.pow(2).mean().sqrt() / l32.double().pow(2).mean().sqrt()),
              "mask_agreement_pct": float((class_mask(l16) == class_mask(l32)).double().mean() * 100.0),
              "loss_f32": s32, "loss_bf16_mixed": s16, "loss_rel_delta": abs(s16 - s32) / abs(s32),
              "grad_cosine": float((g16.double() * g32.double()).sum() / (g16.double().norm() * g32.double().norm())),
              "grad_rel_l2_err": float((g16.double() - g32.double()).norm() / g32.double().norm())}
    if with_oracle:
        try:        # against the fp32 CPU oracle (the reference's arithmetic) on two tiles of the batch
            from oracle import efficientnet_unet_ref as R
            from oracle import losses_ref

            net = R.build(args.version, x.shape[1], 4)
            sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            xs, ys, nz = x[:2].cpu(), y[:2].cpu(), noise[:, :2].cpu()
            with torch.no_grad():
                ref = R.unet_forward(sd, net, xs, training=True, dc_noise=nz)
                ref_loss = float(losses_ref.focal(ref, ys, torch.ones(4), 2.0, 0.0, ignore_index=0))
            bufs = m16._flat_bufs.clone()
            nbt = m16._flat_nbt.clone()
            m16.drop_connect_noise = nz.to(dev)
            with torch.no_grad():
                got = m16(x[:2])
                got_loss = float(loss_fn(got, y[:2]))
            m16._flat_bufs.copy_(bufs)
            m16._flat_nbt.copy_(nbt)
            m16.drop_connect_noise = None
            dd = (got.cpu() - ref).double()
            parity["vs_fp32_cpu_oracle_bs2"] = {
                "logits_max_rel_err": float(dd.abs().max() / ref.abs().max()),
                "logits_rms_rel_err": float(dd.pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()),
                "mask_agreement_pct": float((got.cpu().argmax(1) == ref.argmax(1)).double().mean() * 100.0),
                "loss_oracle": ref_loss, "loss_bf16_mixed": got_loss, "loss_rel_delta": abs(got_loss - ref_loss) / abs(ref_loss)}
        except Exception as e:  # noqa: BLE001
            parity["vs_fp32_cpu_oracle_bs2"] = {"error": repr(e)[:200]}
    # ---- throughput: the headline's protocol ------------------------------------------------------------------------------------
    opt16 = FlatAdam(m16, lr=1.5e-6, weight_decay=0.05)

    def step():
        opt16.zero_grad()
        loss = loss_fn(m16(x), y)
        loss.backward()
        opt16.step()
        return loss

    for _ in range(args.warmup):
        step()
    dt = time_steps(step, args.steps, None, dev)
    value = B * args.steps / dt
    e16 = next(iter(m16._engines.values()))
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(e16.plan.logits_shape, device=dev)
    grads = m16._grad_buffer()
    dout = torch.randn(e16.plan.logits_shape, device=dev) * 1e-6
    kinds, mfma = profile_programs(_lib, D, (e16.fwd, e16.bwd), (e16.bases(m16, x, out, noise=noise),
                                                                 e16.bases(m16, x, None, dout=dout, noise=noise, grads=grads)), st)
    dom = max(mfma, key=lambda k: mfma[k]["ms"])
    k = mfma[dom]
    # the bf16 kernels still read f32 activations: at 32 - 128 flop per byte they sit far left of the bf16 ridge (312 flop per
    # byte), so their roof is HBM; the MFMA figure is given beside it
    roof = {"kernel": dom, "bound": "hbm", "achieved": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"]), "launches": k["launches"],
            "avg_launch_ms": round(k["ms"] / k["launches"], 4),
            "mfma_tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 1), "mfma_peak_bf16_tflops": PEAK_BF16_MFMA_TFLOPS,
            "mfma_frac_of_bf16_peak": round(k["flops"] / (k["ms"] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
            "mfma_kernels": {n: {"ms": round(v["ms"], 3), "launches": v["launches"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                                 "algorithmic_gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                             for n, v in sorted(mfma.items(), key=lambda kv: -kv[1]["ms"])}}
    return {"workload": "the headline workload with dense convs / weight gradients on bf16 MFMA operands (f32 accumulate, f32 BatchNorm "
                        "statistics, loss, master weights, Adam); reported separately from the f32 headline",
            "value": round(value, 2), "unit": "tiles/s", "ms_per_step": round(dt / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
            "dtype": "bf16-mixed", "speedup_vs_f32": round(value / f32_tiles_per_s, 3), "roofline": roof,
            "kernels": {kk: {"ms": round(v["ms"], 3), "launches": v["launches"]} for kk, v in sorted(kinds.items())}, "parity": parity}


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
    ap.add_argument("--no-prithvi", action="store_true", help="skip the extra Prithvi keys (N = 1: MAE and the two segmentation steps; N > 1: the data-parallel MAE step; the headline is unaffected)")
    ap.add_argument("--no-bf16", action="store_true", help="skip the extra key `bf16_mixed` (N = 1 only)")
    ap.add_argument("--mae-batch", type=int, default=64, help="per-GPU batch of the data-parallel Prithvi leg (N > 1 only)")
    ap.add_argument("--mae-steps", type=int, default=5, help="timed steps of the data-parallel Prithvi leg (N > 1 only)")
    ap.add_argument("--sharded-adam", action="store_true", help="N > 1: reduce-scatter the gradient buckets, Adam on the slices each rank owns, all-gather "
                    "the parameters (ddp.FlatGradReducer(mode='sharded')) instead of all-reduce + full Adam on every rank")
    ap.add_argument("--bucket-mb", type=float, default=None, help="gradient bucket size of the data-parallel reducer (default: ddp.DEFAULT_BUCKET_MB = 32)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16-mixed"],
                    help="arithmetic of the HEADLINE run (default f32, the parity path; bf16-mixed is otherwise reported as the extra key `bf16_mixed`)")
    args = ap.parse_args()
    global BUCKET_MB
    BUCKET_MB = args.bucket_mb

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called as the driver calls it (`python bench.py --gpus N ...`): become the launcher; no GPU call before or after
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
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
    model_cfg = EfficientNetConfig(args.version, args.bands, ncls, class_distribution=[0.25] * ncls)
    model = EfficientnetUnet(model_cfg)
    model.to(dev).train()
    model.precision = args.precision
    loss_fn = FocalLoss(torch.ones(ncls), 2.0, 0.0, ignore_index=0)
    ddp = None
    if world > 1:
        from s2lc_amd.ddp import FlatGradReducer

        ddp = FlatGradReducer(model, dist, bucket_mb=BUCKET_MB, mode="sharded" if args.sharded_adam else "allreduce")
        ddp.broadcast_parameters(0)
    opt = FlatAdam(model, lr=1.5e-6, weight_decay=0.05, reducer=ddp)  # BASE_CONFIG lr / weight_decay

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

    last = {}

    def timed_step():
        last["loss"] = step()

    for _ in range(args.warmup):
        step()
    dt = time_steps(timed_step, args.steps, dist, dev)       # EXACTLY args.steps steps, barrier + synchronize on both sides, MAX over ranks
    loss = last["loss"]
    step_ms = dt / args.steps * 1e3

    # ---- outside the timed region: what the all-reduce costs, and N = 1 on the SAME plan ---------------------------------
    allreduce, same_plan = None, None
    if ddp is not None:
        # (a) the same segmented, tape-order backward with the bucket hook replaced by a no-op: no collective at all.  Its rate
        #     is "N = 1 running the program the N > 1 ranks run" (FlatGradReducer switches the deferral of the decoder's weight
        #     gradients off, ddp.py), the fair denominator of a weak-scaling efficiency.
        calls = []
        model._bwd_segment_hook = lambda lo, hi, grads: calls.append((lo, hi))
        opt_reducer, opt.reducer = opt.reducer, None      # (sharded mode: no parameter all-gather either; every rank steps all of its own)
        n_extra = max(3, min(args.steps, 10))
        step()
        calls.clear()
        step()
        buckets = list(calls)
        noop_dt = time_steps(step, n_extra, dist, dev)
        noop_ms = noop_dt / n_extra * 1e3
        same_plan = round(B * n_extra / noop_dt, 2)
        model._bwd_segment_hook = ddp.on_segment
        opt.reducer = opt_reducer
        try:
            allreduce = allreduce_report(model, ddp, dist, dev, world, step_ms, noop_ms, buckets)
        except Exception as e:  # noqa: BLE001
            allreduce = {"error": repr(e)[:300]}
    elif not args.no_profile and rank == 0:
        # N = 1: the default plan defers the decoder's weight gradients (faster alone); also time the tape-order plan that the
        # data-parallel ranks run, so that N > 1 lines can be compared with either
        saved = dict(model._engines)
        try:
            model._defer_wgrads = False
            model._engines.clear()
            model._bwd_segment_hook = lambda lo, hi, grads: None
            n_extra = max(3, min(args.steps, 10))
            step(); step()
            same_plan = round(B * n_extra / time_steps(step, n_extra, None, dev), 2)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: tape-order plan leg failed: {e!r}", file=sys.stderr, flush=True)
        finally:
            model._defer_wgrads = None
            model._bwd_segment_hook = None
            model._engines.clear()
            model._engines.update(saved)        # the default (deferring) plan: the one the timed region ran and the profile leg reads
    loss_val = float(loss.item())

    roofline, kernels, cpu, peaks, adam_ms, extra = None, None, None, None, None, {}
    if rank == 0:
        if not args.no_profile:
            try:
                # once per run: the f32-MFMA issue rate and the stream-copy bandwidth THIS GPU sustains (include/s2k.h s2k_measure_peaks)
                peaks = _lib.measure_peaks(dev)
                adam_ms = time_adam(opt, dev)
                # live per-stage device time (HIP events on the launch stream) of one forward + backward
                eng = next(iter(model._engines.values()))
                st = torch.cuda.current_stream().cuda_stream
                out = torch.empty(eng.plan.logits_shape, device=dev)
                noise = torch.rand(eng.n_noise_rows, B, device=dev)
                grads = model._grad_buffer()
                dout = torch.randn(eng.plan.logits_shape, device=dev) * 1e-6
                kinds, mfma = profile_programs(_lib, D, (eng.fwd, eng.bwd), (eng.bases(model, x, out, noise=noise),
                                                                             eng.bases(model, x, None, dout=dout, noise=noise, grads=grads)), st)
                kernels = {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in sorted(kinds.items())}
                # the committed PMC traffic run measured the headline configuration only
                roofline = make_roofline(mfma, peaks, traffic_workload=(args.version, C, H, B) == ("b5", 13, 256, 32))
            except Exception as e:  # noqa: BLE001   (the measured throughput must still be printed)
                print(f"bench.py: profile leg failed: {e!r}", file=sys.stderr, flush=True)
                roofline = {"error": repr(e)[:300]}
        if world == 1 and args.precision == "f32" and not args.no_bf16 and not args.no_profile:
            try:        # an extra key must never cost the headline line
                extra["bf16_mixed"] = bf16_mixed_leg(model, model_cfg, x, y, loss_fn, args, dev, world * B * args.steps / dt,
                                                     not args.no_cpu_baseline, _lib, D)
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: bf16_mixed failed: {e!r}", file=sys.stderr, flush=True)
                extra["bf16_mixed"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_prithvi and not args.no_profile:
            del model, opt
            torch.cuda.empty_cache()
            for what in ("mae", "seg_frozen", "seg_unfrozen"):
                try:        # an extra key must never cost the headline line
                    extra["prithvi_" + what] = prithvi_workload(what, dev, peaks)
                except Exception as e:  # noqa: BLE001
                    print(f"bench.py: prithvi_{what} failed: {e!r}", file=sys.stderr, flush=True)
                    extra["prithvi_" + what] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            print(f"bench.py: GPU legs done ({world * B * args.steps / dt:.1f} tiles/s); timing the CPU oracle", file=sys.stderr, flush=True)
            try:
                cpu = cpu_baseline(args.version, C, H, ncls)
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: cpu_baseline failed: {e!r}", file=sys.stderr, flush=True)
                cpu = {"error": repr(e)[:300]}
        tiles = world * B * args.steps
        line = {
            "metric": "Sentinel-2 256x256x13 tiles/sec fwd+bwd", "value": round(tiles / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16-mixed", "data": "synthetic",
            "config": {"workload": f"efficientnet-unet-{args.version} {C}x{H}x{H} bs{B}/GPU focal(g=2) train step "
                                   f"(fwd+loss+bwd{'+allreduce' if world > 1 else ''}+adam)",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "roofline": roofline,
            "layerwise_roofline": {"tiles_per_s_per_gpu_at_100pct": UNET_B5_LAYERWISE_TILES_PER_S,
                                   "frac": round(tiles / dt / (world * UNET_B5_LAYERWISE_TILES_PER_S), 4), "source": "SURVEY.md §8d"}
            if (args.version, C, H) == ("b5", 13, 256) else None,
            "cpu_baseline": cpu, "adam_ms": None if adam_ms is None else round(adam_ms, 4),
            "measured_peaks": None if peaks is None else {"mfma_f32_tflops": round(peaks["mfma_f32_tflops"], 1),
                                                          "mfma_clock_mhz": round(peaks["mfma_clock_mhz"]),
                                                          "stream_copy_gbps": round(peaks["copy_gbps"], 1),
                                                          # a fill (write-only) reaches 6.8 TB/s on this part, our read+write copy 5.6-5.9;
                                                          # the guide's own float4 copy measured 6.29 TB/s (MI355X_MICROARCH.md)
                                                          "guide_copy_gbps": GUIDE_COPY_GBS,
                                                          "spec": {"mfma_f32_tflops": PEAK_F32_MFMA_TFLOPS, "hbm_gbps": PEAK_HBM_GBS}},
            "kernels": kernels, "loss": round(loss_val, 6),
            # tiles/s of ONE GPU running the plan the data-parallel ranks run (tape-order weight gradients, segmented backward,
            # no collective): the like-for-like N = 1 of a weak-scaling efficiency
            "n1_same_plan_tiles_per_s": same_plan,
            "allreduce": allreduce, **extra,
        }
    if dist is not None and not args.no_prithvi:
        # N > 1: the data-parallel Prithvi leg (configs[4]).  It is collective, so it runs on every rank after rank 0's profile legs;
        # an extra key must never cost the headline line: a watchdog prints the line without the key and ends the rank if the leg hangs
        import threading

        # ONE place prints the result line, whichever thread gets there first (the timer thread could fire while the main
        # thread was printing - two JSON lines; `line` was mutated without a lock)
        out_lock = threading.Lock()
        state = {"printed": False}

        def print_line(mae_value):
            with out_lock:
                if state["printed"]:
                    return False
                state["printed"] = True
                if rank == 0:
                    line["prithvi_mae"] = mae_value
                    print(json.dumps(line), flush=True)
                return True

        def give_up():
            # the headline measurement is complete and valid: the line goes out with the error in place of the extra key, and the rank
            # ends with 0 (a non-zero code would make the launcher discard a good headline because an EXTRA leg hung); the hang itself
            # is on stderr and in the JSON
            if print_line({"error": f"the data-parallel Prithvi leg did not finish within {MAE_DP_LIMIT_S} s"}):
                print(f"bench.py: rank {rank}: data-parallel Prithvi leg timed out", file=sys.stderr, flush=True)
                os._exit(0)
        dog = threading.Timer(MAE_DP_LIMIT_S, give_up)
        dog.daemon = True
        dist.barrier()          # (the other ranks waited here while rank 0 profiled)
        dog.start()
        ok = 1
        try:
            del model, opt
            torch.cuda.empty_cache()
            mae = prithvi_mae_dp_leg(dist, dev, rank, world, batch=args.mae_batch, steps=args.mae_steps, warmup=2 if args.mae_steps >= 3 else 1)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: prithvi_mae (data-parallel) failed: {e!r}", file=sys.stderr, flush=True)
            mae = {"error": repr(e)[:300]}
            ok = 0
        # did EVERY rank finish the leg?  A rank whose leg raised has left its peers inside the leg's all-reduces: the agreement below
        # (still under the watchdog - it is a collective too) then never completes on the failed rank, the watchdog prints its line
        # and ends it; ranks that do agree on a failure skip the final barrier instead of hanging in it
        all_ok = False
        try:
            t = torch.tensor([ok], device=dev, dtype=torch.int32) if dist.get_backend() == "nccl" else torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            all_ok = bool(int(t.item()))
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: agreement after the Prithvi leg failed: {e!r}", file=sys.stderr, flush=True)
        dog.cancel()
        if not all_ok and "error" not in (mae if isinstance(mae, dict) else {}):
            mae = {"error": "the data-parallel Prithvi leg failed on another rank"}
        print_line(mae)
        if all_ok:
            dist.barrier()
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        return
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()          # rank 0 profiles after the timed region: nobody tears the group down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
