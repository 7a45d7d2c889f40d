"""Flat parameter / gradient / buffer storage.

All parameters of a network live in ONE contiguous fp32 device buffer (and all gradients in
another with the same layout) so that
  * the native stage programs address any parameter as base + offset (include/s2k.h),
  * data-parallel gradient reduction is a handful of large RCCL calls on contiguous slices
    (xGMI likes few, large collectives), and
  * the optimiser can be a single fused pass.
The `nn.Parameter`s the user sees (`state_dict()`, `parameters()`) are views into that buffer,
in the reference's registration order.
"""
from __future__ import annotations

import torch


class FlatParamsMixin:
    _layout = None

    def _init_flat(self, layout) -> None:
        self._layout = layout
        self._engines = {}
        self._flat_grads = None
        self._flatten()

    def _flatten(self) -> None:
        L = self._layout
        named = dict(self.named_parameters())
        if list(named) != list(L.params):
            raise RuntimeError("parameter registration order differs from the planned layout")
        dev = next(iter(named.values())).device
        flat = torch.zeros(L.n_params, dtype=torch.float32, device=dev)
        for name, (off, shape) in L.params.items():
            p = named[name]
            if tuple(p.shape) != tuple(shape):
                raise RuntimeError(f"{name}: shape {tuple(p.shape)} != planned {shape}")
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(shape)
            p.grad = None
        bufs = torch.zeros(max(L.n_bufs, 1), dtype=torch.float32, device=dev)
        nbt = torch.zeros(max(len(L.nbt), 1), dtype=torch.int64, device=dev)
        for name, (off, shape) in L.bufs.items():
            mod_name, attr = name.rsplit(".", 1)
            mod = self.get_submodule(mod_name)
            old = getattr(mod, attr)
            n = old.numel()
            bufs[off:off + n].copy_(old.reshape(-1))
            setattr(mod, attr, bufs[off:off + n].view(shape))
        for i, name in enumerate(L.nbt):
            mod_name, attr = name.rsplit(".", 1)
            mod = self.get_submodule(mod_name)
            nbt[i] = getattr(mod, attr)
            setattr(mod, attr, nbt[i])
        self._flat_params, self._flat_bufs, self._flat_nbt = flat, bufs, nbt
        self._flat_grads = None
        self._engines = {}
        # torch.compile boundary (compile_ops.py): the leaf through which autograd reaches the opaque backward op, and the handle
        # under which the custom ops find this module
        from . import compile_ops

        self._compile_anchor = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)
        self._compile_handle = compile_ops.register(self)

    def __setstate__(self, state):
        """copy.deepcopy / pickle: the copy is a module of its own - its compiled forward must find IT (the handle is id-based and
        would resolve to the original module, or to nothing), and it plans its own engines."""
        super().__setstate__(state)
        from . import compile_ops

        self._engines = {}
        self._compile_handle = compile_ops.register(self)

    # -- arithmetic ---------------------------------------------------------------------------
    @property
    def precision(self) -> str:
        """"f32" (default: exact-f32 MFMA everywhere, the parity path) or "bf16-mixed": dense convs / Linears and their weight
        gradients may round their MFMA operands to bf16 (f32 accumulation, f32 BatchNorm / LayerNorm / attention / loss / master
        weights / Adam) - the arithmetic class of the reference's own default `precision="bf16"` (configs/segmentation.py:146,153,
        prithvi_mae_finetune.py), reported separately from the f32 results and never the default."""
        return getattr(self, "_precision", "f32")

    @precision.setter
    def precision(self, value: str) -> None:
        if value not in ("f32", "bf16-mixed"):
            raise ValueError(f"precision must be 'f32' or 'bf16-mixed', got {value!r}")
        if value != self.precision:
            self._precision = value
            self._engines.clear()

    @property
    def _compile_trainable(self) -> bool:
        return any(p.requires_grad for p in self.parameters())

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self._layout is not None:
            self._flatten()
        return out

    # -- gradients ----------------------------------------------------------------------------
    def _grad_buffer(self) -> torch.Tensor:
        if self._flat_grads is None or self._flat_grads.device != self._flat_params.device:
            self._flat_grads = torch.zeros_like(self._flat_params)
        return self._flat_grads

    def _publish_grads(self, skip: set[str]) -> None:
        """Point every parameter's .grad at its slice of the flat gradient buffer."""
        g = self._flat_grads
        named = dict(self.named_parameters())
        for name, (off, shape) in self._layout.params.items():
            p = named[name]
            if name in skip or not p.requires_grad:
                continue
            n = p.numel()
            p.grad = g[off:off + n].view(shape)

    def _grad_scratch(self) -> torch.Tensor:
        if getattr(self, "_flat_grads2", None) is None or self._flat_grads2.device != self._flat_params.device:
            self._flat_grads2 = torch.zeros_like(self._flat_params)
        return self._flat_grads2

    def _grads_live(self) -> bool:
        """True when the user kept gradients from an earlier backward (accumulation semantics)."""
        if self._flat_grads is None:
            return False
        for p in self.parameters():
            if p.grad is not None:
                return p.grad.data_ptr() >= self._flat_grads.data_ptr() and \
                    p.grad.data_ptr() < self._flat_grads.data_ptr() + self._flat_grads.numel() * 4
        return False

    def _anchor(self, device) -> torch.Tensor:
        a = getattr(self, "_anchor_t", None)
        if a is None or a.device != device:
            a = torch.zeros((), device=device, requires_grad=True)
            self._anchor_t = a
        return a

    @property
    def _no_grad_params(self) -> set:
        return getattr(self, "_unused_params", set())

    def flat_parameters(self) -> torch.Tensor:
        return self._flat_params

    def flat_gradients(self) -> torch.Tensor:
        return self._grad_buffer()
