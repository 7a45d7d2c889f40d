"""CPU, world_size 2 over gloo: the flat-gradient reducer averages gradients like DDP, and the backward
program's segments make every parameter's gradient final before its bucket is reduced."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import s2lc_amd  # noqa: F401
from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
from s2lc_amd.plan import opdefs as D
from s2lc_amd.plan.program import TRef


def test_backward_segments_cover_every_parameter_after_its_last_write():
    m = EfficientnetUnet(EfficientNetConfig("b0", 6, 4, class_distribution=[.25] * 4))
    plan = m._make_plan(2, 64, 64, True, ) if False else __import__("s2lc_amd.plan.unet_plan", fromlist=["plan_unet"]).plan_unet(
        m.spec, 2, 64, 64, True, m._layout, bucket_floats=1 << 20)
    segs = plan.bwd_param_marks
    assert len(segs) >= 3
    # contiguous op ranges covering the whole program, contiguous descending buckets covering all floats
    assert segs[0][0] == 0 and segs[-1][1] == len(plan.bwd)
    for (a0, b0, lo0, hi0), (a1, b1, lo1, hi1) in zip(segs, segs[1:]):
        assert b0 == a1 and lo0 == hi1
    assert segs[0][3] == plan.layout.n_params and segs[-1][2] == 0
    # no op after a segment's end writes into that segment's bucket (grads or weight-grad scratch)
    for (a, b, lo, hi) in segs:
        for kind, fields in plan.bwd.ops[b:]:
            for v in fields.values():
                if isinstance(v, TRef) and v.base in (D.BASE["GRADS"], D.BASE["WGS"]) and kind not in ("MEMSET", "WGRAD_FINALIZE"):
                    assert not (lo <= v.off // 4 < hi), (kind, v.name)
    # every conv weight is folded exactly once
    folded = 0
    for kind, fields in plan.bwd.ops:
        if kind == "WGRAD_FINALIZE":
            folded += fields["N_ENTRIES"]
    # (1x1 weights accumulate straight into the gradient buffer: only the scratch-bound ones are folded)
    assert folded == sum(1 for k in plan.bwd.ops if k[0] == "WGRAD" and k[1]["WGS"].base == D.BASE["WGS"] and k[1]["WGS"].off // 4 in
                         {off for off, _ in plan.layout.params.values()})
    assert any(k[0] == "WGRAD" and k[1]["WGS"].base == D.BASE["GRADS"] for k in plan.bwd.ops)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from s2lc_amd.ddp import FlatGradReducer

        torch.manual_seed(100 + rank)  # different initial weights: broadcast must fix that
        m = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[.25] * 4))
        red = FlatGradReducer(m, dist)
        red.broadcast_parameters(0)
        w0 = m._flat_params.clone()
        assert m._grad_scale == 0.5
        grads = m._grad_buffer()
        g = torch.Generator().manual_seed(rank)
        local = torch.randn(grads.shape, generator=g)
        # what the engine does: local gradients already carry the 1/world factor, buckets arrive back to front
        grads.copy_(local * m._grad_scale)
        n = grads.numel()
        cuts = [n, 2 * n // 3, n // 3, 0]
        for hi, lo in zip(cuts, cuts[1:]):
            red.on_segment(lo, hi, grads)
        red.finish()
        # the separately callable methods (vit_engine._MethodFunction) add local, 1/world-scaled gradients to the flat buffer
        # and only mark the module: finish() must then reduce the whole buffer itself, exactly once
        local2 = torch.randn(grads.shape, generator=g)
        grads2 = local2 * m._grad_scale
        m._flat_grads = grads2
        m._method_grads_unreduced = True
        red.finish()
        assert not m._method_grads_unreduced
        red.finish()            # nothing pending, nothing marked: must not reduce again
        torch.save((rank, grads.clone(), local, w0, grads2.clone(), local2), os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_flat_grad_reducer_averages_like_ddp_gloo(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    out = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    out.sort(key=lambda t: t[0])
    mean = (out[0][2] + out[1][2]) / 2
    mean2 = (out[0][5] + out[1][5]) / 2
    for rank, reduced, _, w0, reduced2, _ in out:
        assert torch.allclose(reduced, mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(reduced2, mean2, rtol=1e-6, atol=1e-7)
    assert torch.equal(out[0][3], out[1][3])  # identical weights after the broadcast


def _sharded_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from s2lc_amd.ddp import FlatGradReducer

        torch.manual_seed(7)
        m = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[.25] * 4))
        red = FlatGradReducer(m, dist, mode="sharded")
        grads = m._grad_buffer()
        local = torch.randn(grads.shape, generator=torch.Generator().manual_seed(rank))
        grads.copy_(local * m._grad_scale)
        n = grads.numel()
        cuts = [n, 2 * n // 3 // 64 * 64 + 40, n // 3 // 64 * 64, 0]       # one bucket boundary off the 64-float grid: leftovers
        for hi, lo in zip(cuts, cuts[1:]):
            red.on_segment(lo, hi, grads)
        red.finish()
        owned = red.owned(0, n)
        # "parameters": every rank writes its rank id + 1 into what it owns, the all-gather must deliver everyone's slices
        flat = torch.zeros(n)
        for a, b in owned:
            flat[a:b] = rank + 1.0
        red.all_gather_slices(flat)
        torch.save((rank, grads.clone(), local, owned, flat, list(red.last_segments), red._native_rs), os.path.join(outdir, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sharded_reducer_slices_gloo(tmp_path):
    """mode="sharded" (reduce-scatter per bucket): every rank holds the rank-mean of the gradients on the slices it owns, the slices of
    a bucket tile it together with the replicated leftover, and all_gather_slices delivers every owner's values to every rank."""
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    out = sorted((torch.load(tmp_path / f"s{r}.pt") for r in range(world)), key=lambda t: t[0])
    mean = (out[0][2] + out[1][2]) / 2
    n = mean.numel()
    cover = torch.zeros(n)
    for rank, grads, _, owned, flat, segs, native in out:
        assert len(segs) == 3 and all(sl % 64 == 0 for _, _, sl in segs)
        for a, b in owned:
            assert torch.allclose(grads[a:b], mean[a:b], rtol=1e-6, atol=1e-7), (rank, a, b)
            cover[a:b] += 1
    leftovers = torch.zeros(n)
    for lo, hi, sl in out[0][5]:
        leftovers[lo + sl * world: hi] = 1
    assert leftovers.sum() > 0, "the test must exercise a bucket with a leftover"
    assert torch.equal(cover, 1 + leftovers * (world - 1)), "slices tile every bucket once; leftovers belong to every rank"
    # after the all-gather: a slice carries its owner's id on every rank; leftovers each rank's own
    for rank, _, _, _, flat, segs, _ in out:
        for lo, hi, sl in segs:
            for r in range(world):
                assert (flat[lo + r * sl: lo + (r + 1) * sl] == r + 1.0).all()
            assert (flat[lo + sl * world: hi] == rank + 1.0).all()
    print(f"sharded reducer over gloo: native reduce-scatter {out[0][6]}")


def test_run_backward_modes_without_a_gpu():
    """engine.run_backward, the one place that decides how a backward program meets the reducer (eager nodes and compiled ops):
    no reducer -> one run; reducer -> segments + hook; no_sync -> one run, marked for finish(); accumulate -> one run, marked;
    accumulate onto bucket-reduced gradients -> refused."""
    from s2lc_amd.engine import run_backward

    class M:
        pass

    marks = [(0, 2, 8, 16), (2, 5, 0, 8)]
    m, ran = M(), []
    run_backward(m, marks, 5, lambda a, b: ran.append((a, b)), None, accumulate=False)
    assert ran == [(0, 5)] and not getattr(m, "_bucket_reduced", False)
    m, ran, hooked = M(), [], []
    m._bwd_segment_hook = lambda lo, hi, g: hooked.append((lo, hi))
    run_backward(m, marks, 5, lambda a, b: ran.append((a, b)), None, accumulate=False, lo_min=4)
    assert ran == [(0, 2), (2, 5)] and hooked == [(8, 16), (4, 8)] and m._bucket_reduced
    m, ran, hooked = M(), [], []
    m._bwd_segment_hook = lambda lo, hi, g: hooked.append((lo, hi))
    m._no_sync = True
    run_backward(m, marks, 5, lambda a, b: ran.append((a, b)), None, accumulate=False)
    assert ran == [(0, 5)] and hooked == [] and m._method_grads_unreduced
    m._no_sync = False
    run_backward(m, marks, 5, lambda a, b: ran.append((a, b)), None, accumulate=True)      # the last micro-batch
    assert ran == [(0, 5), (0, 5)] and hooked == [] and m._method_grads_unreduced
    m2 = M()
    m2._bwd_segment_hook = lambda lo, hi, g: None
    m2._bucket_reduced = True
    with pytest.raises(RuntimeError, match="no_sync"):
        run_backward(m2, marks, 5, lambda a, b: None, None, accumulate=True)


def test_reducer_bucket_size_is_a_planner_parameter():
    """FlatGradReducer(bucket_mb=...) reaches the planner: smaller buckets -> more backward segments, same coverage"""
    from s2lc_amd.plan.unet_plan import plan_unet

    m = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[.25] * 4))
    n = [len(plan_unet(m.spec, 2, 64, 64, True, m._layout, bucket_floats=int(mb * (1 << 20)) // 4).bwd_param_marks) for mb in (32.0, 4.0, 1.0)]
    assert n[0] <= n[1] <= n[2] and n[2] > n[0]
