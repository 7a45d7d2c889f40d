"""Data-parallel gradient reduction over RCCL / xGMI.

The reference relies on Lightning's implicit DDP (train_segmentation.py:273-280): NCCL all-reduce of
25 MB buckets launched from autograd hooks.  Here gradients already live in ONE flat buffer and the
backward program is cut into a few segments after which a contiguous suffix of that buffer is final
(plan.bwd_param_marks), so each rank issues a handful of large all-reduces on a side stream while the
next backward segment is still running: few, large collectives suit the point-to-point xGMI mesh
(7 links per GPU), and nothing on the data path is per-tensor.

Semantics match DDP: every rank's loss is a per-rank mean, gradients are averaged over ranks (the 1/N
is folded into the upstream gradient, so the collective is a plain SUM), BatchNorm statistics stay
per-rank (the reference never enables SyncBatchNorm).

Two collectives (SURVEY.md §8e names both):
  * mode="allreduce" (default): one all-reduce per bucket; every rank then runs Adam over all parameters.
  * mode="sharded": the two halves of that all-reduce with the optimiser between them - each bucket is REDUCE-SCATTERed
    (rank r receives the summed gradients of the r-th slice of the bucket), `FlatAdam(module, reducer=...)` updates only the
    slices this rank owns (1/N of the Adam work per GPU), and the updated PARAMETERS of each bucket are ALL-GATHERed.  Same
    bytes on the links as the all-reduce; what moves is where the second half sits: behind the optimiser instead of behind the
    backward, so it is exposed unless the next step's first layers are cheap - measure before preferring it (bench.py
    --sharded-adam reports the step time either way).  Outside the slices a rank owns, `.grad` holds that rank's LOCAL
    gradients after a step in this mode.
"""
from __future__ import annotations

import contextlib

import torch

# Bucket size (floats of the flat gradient buffer per all-reduce).  SURVEY.md 8e's arithmetic for the b5 U-Net: 40.3 M parameters =
# 161 MB of gradients per step; a GPU has 7 xGMI links of ~153 GB/s, and RCCL's all-reduce over the fully connected mesh moves
# 2 (N - 1) / N x bytes per GPU, so at the ~300 GB/s bus bandwidth a ring of 8 reaches the whole buffer costs ~0.95 ms of a 21 ms
# backward - what matters is not bandwidth but (a) the fixed cost per collective (30 - 50 us each over 8 ranks: dozens of 25 MB
# DDP-style buckets from hooks would cost more in launches than in bytes; 5 buckets cost 0.2 ms) and (b) the size of the LAST
# bucket, which is the only one nothing hides: buckets close from the END of the flat buffer (the decoder, whose gradients are
# final first) towards its start, so the last bucket holds the stem and the first encoder blocks - a few hundred KB in this
# network whatever the nominal size.  32 MB (8 M floats) gives 5 buckets for the b5 U-Net and 13 for Prithvi-100M; the planner
# takes any value (`FlatGradReducer(..., bucket_mb=...)`), bench.py reports the sizes that resulted in `allreduce.bucket_bytes`.
DEFAULT_BUCKET_MB = 32.0


class FlatGradReducer:
    def __init__(self, module, dist, process_group=None, bucket_mb: float | None = None, mode: str = "allreduce"):
        if mode not in ("allreduce", "sharded"):
            raise ValueError("mode is 'allreduce' or 'sharded'")
        self.module, self.dist, self.group, self.mode = module, dist, process_group, mode
        self.bucket_mb = float(DEFAULT_BUCKET_MB if bucket_mb is None else bucket_mb)
        if self.bucket_mb <= 0:
            raise ValueError("bucket_mb must be positive")
        if bucket_mb is not None or not hasattr(module, "_bucket_floats"):
            module._bucket_floats = max(1, int(self.bucket_mb * (1 << 20)) // 4)     # read by the planners (plan/unet_plan.py, vit_plan.py)
        else:
            self.bucket_mb = module._bucket_floats * 4 / (1 << 20)                      # set on the module beforehand
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.segments = []          # sharded: [(lo, hi, slice floats)] of the step being reduced, in the order the buckets closed
        self.last_segments = []     # ... of the last finished step (what FlatAdam steps over)
        self._shard_buf, self._shard_used = None, 0
        self._native_rs = None      # does the backend reduce-scatter / all-gather tensors of this device itself? (probed once)
        self.on_gpu = module._flat_params.is_cuda
        self.stream = torch.cuda.Stream(device=module._flat_params.device) if self.on_gpu else None
        self.pending = []
        # Single-GPU plans issue the decoder's large weight gradients late (they overlap the encoder's HBM-bound backward), which
        # makes every gradient bucket final only at the end of the backward.  With a reducer attached the buckets should become
        # final progressively instead, so that all but the last all-reduce hide behind the remaining backward (over xGMI the
        # 161 MB of a b5 are 1-2.5 ms exposed otherwise, more than the deferral gains): plans built from now on keep tape order.
        module._defer_wgrads = False
        if getattr(module, "_engines", None):
            module._engines.clear()
        module._grad_scale = 1.0 / self.world        # folded into dlogits by the engine
        module._bwd_segment_hook = self.on_segment    # called after each backward segment is enqueued

    def on_segment(self, lo: int, hi: int, grads: torch.Tensor) -> None:
        """Gradients of flat floats [lo, hi) are final on the current stream: reduce them."""
        if hi <= lo:
            return
        view = grads[lo:hi]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(grads.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._reduce(lo, hi, grads, view)
        else:
            self._reduce(lo, hi, grads, view)

    def _reduce(self, lo: int, hi: int, grads: torch.Tensor, view: torch.Tensor) -> None:
        d, SUM = self.dist, self.dist.ReduceOp.SUM
        if self.mode == "allreduce":
            self.pending.append((d.all_reduce(view, op=SUM, group=self.group, async_op=True), None, None))
            return
        # sharded: equal slices of 64-float granularity per rank; what is left over at the end of the bucket (< 64 * world floats) is
        # all-reduced and belongs to every rank
        sl = (hi - lo) // (64 * self.world) * 64
        main = sl * self.world
        self.segments.append((lo, hi, sl))
        if sl:
            own = grads[lo + self.rank * sl: lo + (self.rank + 1) * sl]
            if self._native():
                if self._shard_buf is None or self._shard_buf.device != grads.device:
                    self._shard_buf = torch.empty(grads.numel() // self.world + 64, device=grads.device, dtype=grads.dtype)
                if self._shard_used + sl > self._shard_buf.numel():       # (a second pass over the buffer within one step: accumulation)
                    self._shard_used = 0
                out = self._shard_buf[self._shard_used: self._shard_used + sl]
                self._shard_used += sl
                self.pending.append((d.reduce_scatter_tensor(out, grads[lo: lo + main], op=SUM, group=self.group, async_op=True), out, own))
            else:       # a backend without reduce-scatter for this device (gloo rehearsals on one GPU): the all-reduce leaves the slice in place
                self.pending.append((d.all_reduce(grads[lo: lo + main], op=SUM, group=self.group, async_op=True), None, None))
        if main < hi - lo:
            self.pending.append((d.all_reduce(grads[lo + main: hi], op=SUM, group=self.group, async_op=True), None, None))

    def _native(self) -> bool:
        if self._native_rs is None:
            dev = self.module._flat_params.device
            try:
                a = torch.zeros(64 * self.world, device=dev)
                b = torch.zeros(64, device=dev)
                self.dist.reduce_scatter_tensor(b, a, group=self.group)
                self.dist.all_gather_into_tensor(a, b, group=self.group)
                self._native_rs = True
            except Exception:       # noqa: BLE001 - "not supported by this backend / device" comes in several exception types
                self._native_rs = False
            flag = torch.tensor([1.0 if self._native_rs else 0.0], device=dev)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)      # all ranks take the same path
            self._native_rs = bool(flag.item() > 0.5)
        return self._native_rs

    def all_gather_slices(self, flat: torch.Tensor, segments=None) -> None:
        """sharded mode: every rank's slices of `flat` (parameters after the optimiser, or Adam moments before a checkpoint) to all
        ranks, bucket by bucket of the last finished step (or of the given bucket list)."""
        for lo, hi, sl in (self.last_segments if segments is None else segments):
            if not sl:
                continue
            main = sl * self.world
            if self._native():
                own = flat[lo + self.rank * sl: lo + (self.rank + 1) * sl].clone()
                self.dist.all_gather_into_tensor(flat[lo: lo + main], own, group=self.group)
            else:
                for r in range(self.world):
                    self.dist.broadcast(flat[lo + r * sl: lo + (r + 1) * sl], self._global_rank(r), group=self.group)

    def _global_rank(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def owned(self, a: int, b: int, segments=None):
        """sharded mode: the parts of flat floats [a, b) this rank updates - its slice of every bucket, each bucket's leftover, and
        anything no bucket of the last step covered (replicated)."""
        pieces, covered = [], []
        for lo, hi, sl in (self.last_segments if segments is None else segments):
            main = sl * self.world
            for x, y in ((lo + self.rank * sl, lo + (self.rank + 1) * sl), (lo + main, hi)):
                x, y = max(x, a), min(y, b)
                if y > x:
                    pieces.append((x, y))
            covered.append((lo, hi))
        pos = a
        for lo, hi in sorted(covered):
            if lo > pos:
                pieces.append((pos, min(lo, b)))
            pos = max(pos, hi)
            if pos >= b:
                break
        if pos < b:
            pieces.append((pos, b))
        return sorted(p for p in pieces if p[1] > p[0])

    def finish(self) -> None:
        """Make the reduced gradients visible to the compute stream (call before the optimiser)."""
        mod = self.module
        if getattr(mod, "_method_grads_unreduced", False):
            # gradients from separately called methods (MaskedAutoencoderViT.forward_encoder / _decoder / _loss,
            # EfficientNet.encode / forward): several autograd nodes added their local, already 1/world-scaled gradients to the
            # flat buffer; nothing was final before the last of them, so the trainable range is reduced here in one collective
            # (frozen / never-used ranges hold zeros and simply ride along)
            grads = mod._grad_buffer()
            self.on_segment(0, grads.numel(), grads)
            mod._method_grads_unreduced = False
        mod._bucket_reduced = False
        with (torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()):
            for w, out, own in self.pending:
                w.wait()
                if out is not None:
                    own.copy_(out)       # the reduced slice goes where the optimiser reads gradients
        self.pending.clear()
        self.last_segments, self.segments, self._shard_used = self.segments or self.last_segments, [], 0
        if self.on_gpu:
            torch.cuda.current_stream(self.module._flat_params.device).wait_stream(self.stream)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (torch DDP's `no_sync`): backwards inside the context add LOCAL gradients to the flat buffer and
        start no collective; the first backward outside it adds its own and `finish()` then all-reduces the accumulated sum once
        (one collective over the trainable range - the bucket overlap is given up for that step, the bytes are the same)."""
        mod = self.module
        prev = getattr(mod, "_no_sync", False)
        mod._no_sync = True
        try:
            yield
        finally:
            mod._no_sync = prev

    def broadcast_parameters(self, src: int = 0) -> None:
        """Start from identical weights / BatchNorm buffers on every rank (DDP does this at construction)."""
        self.dist.broadcast(self.module._flat_params, src, group=self.group)
        self.dist.broadcast(self.module._flat_bufs, src, group=self.group)
