"""Fused Adam over the flat parameter buffer.

Same update as the reference's `torch.optim.Adam(lr, weight_decay)` (L2-coupled decay, betas
(0.9, 0.999), eps 1e-8; /root/reference/src/train_segmentation.py:109-127) but one HIP launch per
contiguous parameter range instead of ~900 per-tensor updates.  Parameters that never receive a
gradient (`encoder.fc.*`) are skipped entirely, like torch skips `p.grad is None`.
"""
from __future__ import annotations

import torch

from . import _lib


class FlatAdam:
    def __init__(self, module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.module = module
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.m = None
        self.v = None
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": betas, "eps": eps}]

    def _ranges(self):
        """Contiguous float ranges of the flat buffer that take part in the update."""
        L = self.module._layout
        skip = self.module._no_grad_params
        ranges, start, end = [], None, None
        for name, (off, shape) in L.params.items():
            n = 1
            for s in shape:
                n *= s
            if name in skip:
                if start is not None:
                    ranges.append((start, end))
                    start = None
                continue
            if start is None:
                start = off
            end = off + n
        if start is not None:
            ranges.append((start, end))
        return ranges

    def zero_grad(self, set_to_none: bool = True) -> None:
        """O(1): the next backward overwrites the flat gradient buffer instead of accumulating;
        the per-parameter `.grad` views stay in place (no 900-tensor Python loop per step)."""
        self.module._overwrite_next = True

    @torch.no_grad()
    def step(self) -> None:
        mod = self.module
        p, g = mod._flat_params, mod._grad_buffer()
        if not p.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only")
        if self.m is None or self.m.device != p.device:
            self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.step_count += 1
        grp = self.param_groups[0]
        st = torch.cuda.current_stream(p.device).cuda_stream
        L = _lib.lib()
        for a, b in self._ranges():
            _lib.check(L.s2k_adam_step(p.data_ptr() + 4 * a, g.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a,
                                       self.v.data_ptr() + 4 * a, b - a, grp["lr"], self.betas[0], self.betas[1],
                                       self.eps, grp["weight_decay"], self.step_count, st))
