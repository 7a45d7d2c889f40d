"""Fused Adam over the flat parameter buffer.

Same update as the reference's `torch.optim.Adam(lr, weight_decay)` (L2-coupled decay, betas
(0.9, 0.999), eps 1e-8; /root/reference/src/train_segmentation.py:109-127, train_mae_prithvi.py:98-116)
but one HIP launch per contiguous parameter range instead of ~900 per-tensor updates.  Like torch, a
parameter takes part in a step iff it `requires_grad` and holds a gradient (`p.grad is not None`):
never-used parameters (`encoder.fc.*`), the fixed position tables and anything the user froze after
construction are skipped entirely — no moment update, no weight decay.

Data-parallel option (ddp.FlatGradReducer(mode="sharded")): `FlatAdam(module, reducer=red)` updates only the slices of the flat
buffer this rank received reduced gradients for and all-gathers the updated parameters (see ddp.py); moments of the other ranks'
slices stay zero here until `consolidate_state()` gathers them for a checkpoint.

It IS a `torch.optim.Optimizer` (param_groups, state_dict / load_state_dict, add-on lr schedulers such
as the reference's `get_lr_scheduler` wrap it unchanged); only `step` / `zero_grad` are replaced.
"""
from __future__ import annotations

import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, reducer=None):
        if not hasattr(module, "_flat_params"):
            raise TypeError("FlatAdam optimises a module that keeps its parameters in one flat buffer (flat.FlatParamsMixin)")
        if reducer is not None and reducer.module is not module:
            raise ValueError("the reducer belongs to another module")
        self.module = module
        self.reducer = reducer if (reducer is not None and reducer.mode == "sharded") else None
        self._consolidated = True       # sharded: the moments of every rank's slices are present on this rank
        self._seg_layout = None         # sharded: the bucket list the moments are currently sliced by
        self.step_count = 0
        self.m = None
        self.v = None
        self._plist = None
        self._resumed_without_first = False      # a state dict written before per-parameter steps existed: every active parameter began at step 1
        self._legacy_first = None                # "first_step" of a round-3 state dict (load_state_dict)
        self._updates = {}          # parameter index -> number of steps in which it was UPDATED (torch keeps state['step'] per parameter
                                    # and advances it only when the parameter holds a gradient: freeze / unfreeze / refreeze schedules)
        super().__init__([p for p in module.parameters()], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    def add_param_group(self, param_group) -> None:
        """One group: the fused step runs over contiguous ranges of ONE flat buffer with one set of hyper-parameters (the reference
        configures `torch.optim.Adam(net.parameters(), lr, weight_decay)`, a single group: train_segmentation.py:109-115)."""
        if len(self.param_groups) >= 1:
            raise ValueError("FlatAdam holds one parameter group (one lr / betas / eps / weight_decay for the module's flat buffer); "
                             "per-group hyper-parameters are not supported - use torch.optim.Adam on module.parameters() for that")
        super().add_param_group(param_group)

    # -- which floats take part --------------------------------------------------------------------------
    def _ranges(self):
        """[(first float, end float, local step)]: contiguous ranges of the flat buffer whose parameters currently hold a gradient
        and share a step count.  Like torch.optim.Adam (state['step'] per parameter), a parameter's bias correction counts the
        steps IT has taken: a backbone unfrozen after N steps starts at step 1, not N + 1."""
        mod = self.module
        if self._plist is None:
            named = dict(mod.named_parameters())
            skip = mod._no_grad_params
            # every tensor starts on a 64-float boundary of the flat buffer; the padding floats (zeros, zero gradient) belong
            # to the tensor in front of them, so neighbouring tensors form one contiguous range
            self._plist = [(named[name], off, (named[name].numel() + 63) // 64 * 64, name in skip)
                           for name, (off, shape) in mod._layout.params.items()]
        ranges, start, end, cur = [], None, None, None
        for i, (p, off, n, skipped) in enumerate(self._plist):
            if skipped or not p.requires_grad or p.grad is None:
                if start is not None:
                    ranges.append((start, end, cur))
                    start = None
                continue
            if self._legacy_first is not None:      # state written before the per-parameter update count existed (see load_state_dict)
                first = self._legacy_first.get(i, 1 if self._resumed_without_first else self.step_count)
                self._updates.setdefault(i, self.step_count - first)
            local = self._updates.get(i, 0) + 1
            self._updates[i] = local
            if start is not None and local != cur:
                ranges.append((start, end, cur))
                start = None
            if start is None:
                start, cur = off, local
            end = off + n
        if start is not None:
            ranges.append((start, end, cur))
        self._resumed_without_first = False
        self._legacy_first = None
        return ranges

    def zero_grad(self, set_to_none: bool = True) -> None:
        """O(1): the next backward overwrites the flat gradient buffer instead of accumulating;
        the per-parameter `.grad` views stay in place (no 900-tensor Python loop per step)."""
        self.module._overwrite_next = True

    @torch.no_grad()
    def step(self, closure=None, gather: bool = True):
        """gather=False (sharded mode, measurement only): update the owned slices without the collectives of a step - no parameter
        all-gather, no re-slicing of the moments; the ranks' weights then differ until the next full step."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        mod = self.module
        p, g = mod._flat_params, mod._grad_buffer()
        if not p.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only")
        if self.m is None or self.m.device != p.device:
            self.m = torch.zeros_like(p) if self.m is None else self.m.to(p.device)
            self.v = torch.zeros_like(p) if self.v is None else self.v.to(p.device)
        self.step_count += 1
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        st = torch.cuda.current_stream(p.device).cuda_stream
        L = _lib.lib()
        red = self.reducer
        if red is not None and gather:
            layout = list(red.last_segments)
            if self._seg_layout is not None and layout != self._seg_layout:
                # the buckets changed (another plan, or an accumulated step reduced as one range): slices change owners, so every rank
                # first needs the moments of the slices it is about to own - all ranks see the same change in the same step
                self.consolidate_state()
            self._seg_layout = layout
        with torch.cuda.device(p.device):
            for a0, b0, local_step in self._ranges():
                for a, b in (red.owned(a0, b0) if red is not None else ((a0, b0),)):
                    _lib.check(L.s2k_adam_step(p.data_ptr() + 4 * a, g.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a,
                                               self.v.data_ptr() + 4 * a, b - a, float(grp["lr"]), float(b1), float(b2),
                                               float(grp["eps"]), float(grp["weight_decay"]), local_step, st))
            if red is not None:
                if gather:
                    red.all_gather_slices(p)        # every rank's updated slices to all ranks, bucket by bucket
                self._consolidated = False
        return loss

    def consolidate_state(self) -> None:
        """sharded mode, a COLLECTIVE (call on every rank, e.g. before a checkpoint): gathers the Adam moments of all ranks' slices, after
        which `state_dict()` is the same full state on every rank as the all-reduce mode would hold."""
        if self.reducer is not None and not self._consolidated and self.m is not None and self._seg_layout is not None:
            self.reducer.all_gather_slices(self.m, self._seg_layout)
            self.reducer.all_gather_slices(self.v, self._seg_layout)
        self._consolidated = True

    # -- checkpointing -------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        if self.reducer is not None and not self._consolidated:
            raise RuntimeError("FlatAdam in sharded mode holds only this rank's slices of the moments: call consolidate_state() on "
                               "EVERY rank first (a collective), then state_dict() where the checkpoint is written")
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"state": {"step": self.step_count, "updates": dict(self._updates),
                          "exp_avg": None if self.m is None else self.m.detach().clone(),
                          "exp_avg_sq": None if self.v is None else self.v.detach().clone()},
                "param_groups": groups, "layout_floats": int(self.module._flat_params.numel())}

    def load_state_dict(self, state: dict) -> None:
        if int(state.get("layout_floats", -1)) != int(self.module._flat_params.numel()):
            raise ValueError("optimizer state belongs to a module with a different flat parameter layout")
        st = state["state"]
        self.step_count = int(st["step"])
        self._consolidated, self._seg_layout = True, None       # a checkpoint holds the full moments
        # "updates": per-parameter count of steps taken (torch's state['step']).  Older checkpoints carry "first_step" (the global step
        # at which a parameter first held a gradient: exact unless it was frozen again in between) or nothing (every parameter that
        # holds a gradient at the first resumed step is assumed to have been updated in every step so far)
        self._updates = {int(k): int(v) for k, v in st.get("updates", {}).items()}
        self._legacy_first = None
        self._resumed_without_first = False
        if "updates" not in st:
            self._legacy_first = {int(k): int(v) for k, v in st.get("first_step", {}).items()}
            self._resumed_without_first = "first_step" not in st
        dev = self.module._flat_params.device
        self.m = None if st["exp_avg"] is None else st["exp_avg"].to(device=dev, dtype=torch.float32).clone()
        self.v = None if st["exp_avg_sq"] is None else st["exp_avg_sq"].to(device=dev, dtype=torch.float32).clone()
        for g, saved in zip(self.param_groups, state["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
