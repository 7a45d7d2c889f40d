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
"""
from __future__ import annotations

import torch


class FlatGradReducer:
    def __init__(self, module, dist, process_group=None):
        self.module, self.dist, self.group = module, dist, process_group
        self.world = dist.get_world_size(process_group)
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
                self.pending.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

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
        for w in self.pending:
            w.wait()
        self.pending.clear()
        if self.on_gpu:
            torch.cuda.current_stream(self.module._flat_params.device).wait_stream(self.stream)

    def broadcast_parameters(self, src: int = 0) -> None:
        """Start from identical weights / BatchNorm buffers on every rank (DDP does this at construction)."""
        self.dist.broadcast(self.module._flat_params, src, group=self.group)
        self.dist.broadcast(self.module._flat_bufs, src, group=self.group)
