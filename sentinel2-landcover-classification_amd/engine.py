"""Runtime glue: owns the device buffers of a planned network and runs its stage programs.

PyTorch supplies device memory, the current HIP stream and the autograd *edge* (one
`autograd.Function` for the whole network: forward = one `s2k_program_run`, backward = one more).
The reference's equivalent is `self.net(x)` + `loss.backward()` in
/root/reference/src/train_segmentation.py:129-147 dispatching ~1,000 ATen ops per step.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class UnetEngine:
    """Buffers + packed programs for one (B, H, W, training) shape of one module."""

    def __init__(self, module, B: int, H: int, W: int, training: bool, device: torch.device):
        plan = module._make_plan(B, H, W, training)
        self.plan = plan
        self.fwd = plan.fwd.pack()
        self.bwd = plan.bwd.pack() if plan.bwd is not None else None
        self.device = device
        pad = 256
        self.ws = torch.empty(plan.ws_bytes + pad, dtype=torch.uint8, device=device)
        self.aux = torch.zeros(max(plan.aux_bytes, 8) + pad, dtype=torch.uint8, device=device)
        self.const = torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32, device=device)
        self.wpack = torch.zeros(plan.wpack_bytes // 4 + 65536, dtype=torch.float32, device=device)  # + slack: A-tile loads may overrun
        self.wgs = torch.empty(plan.layout.n_params, dtype=torch.float32, device=device) if training else None
        self.n_noise_rows = plan.n_noise_rows
        self.B = B
        self.bwd_marks = plan.bwd_param_marks

    def bases(self, module, x, out, dout=None, noise=None, grads=None) -> _lib.Bases:
        b = _lib.Bases()
        b.set("WS", self.ws).set("AUX", self.aux).set("CONST", self.const).set("WPACK", self.wpack)
        b.set("PARAMS", module._flat_params).set("BUFS", module._flat_bufs)
        b.set("X", x).set("OUT", out)
        if self.wgs is not None:
            b.set("WGS", self.wgs)
        if dout is not None:
            b.set("DOUT", dout)
        if noise is not None:
            b.set("NOISE", noise)
        if grads is not None:
            b.set("GRADS", grads)
        return b


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _engine(module, x: torch.Tensor, training: bool) -> UnetEngine:
    key = (tuple(x.shape), training, x.device)
    eng = module._engines.get(key)
    if eng is None:
        B, _, H, W = x.shape
        eng = UnetEngine(module, B, H, W, training, x.device)
        module._engines[key] = eng
    return eng


class _UnetFunction(torch.autograd.Function):
    """Whole-network autograd node.  Parameter gradients are written straight into the module's
    flat gradient buffer by the backward program (never returned through autograd: 900+
    per-tensor accumulations would cost more host time than the GPU step)."""

    @staticmethod
    def forward(ctx, x, anchor, module, eng, noise):
        out = torch.empty(eng.plan.logits_shape, dtype=torch.float32, device=x.device)
        bases = eng.bases(module, x, out, noise=noise)
        _lib.run(eng.fwd, bases, _stream(x.device))
        ctx.module, ctx.eng, ctx.noise = module, eng, noise
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        module, eng = ctx.module, ctx.eng
        (x,) = ctx.saved_tensors
        dout = dout.contiguous()
        scale = getattr(module, "_grad_scale", 1.0)
        if scale != 1.0:
            dout = dout * scale   # data-parallel mean folded into the upstream gradient
        live = module._grads_live()
        accumulate = live and not getattr(module, "_overwrite_next", False)
        module._overwrite_next = False
        grads = module._grad_buffer() if not accumulate else module._grad_scratch()
        grads.zero_()
        bases = eng.bases(module, x, None, dout=dout, noise=ctx.noise, grads=grads)
        hook = getattr(module, "_bwd_segment_hook", None)
        st = _stream(x.device)
        if hook is None:
            _lib.run(eng.bwd, bases, st)
        else:
            if accumulate:
                raise RuntimeError("gradient accumulation together with the data-parallel reducer is not supported")
            for (a, b, lo, hi) in eng.bwd_marks:
                _lib.run(eng.bwd, bases, st, a, b)
                hook(lo, hi, grads)
        if accumulate:
            module._grad_buffer().add_(grads)
        if not live:
            module._publish_grads(module._no_grad_params)
        return None, None, None, None, None


def run_unet(module, x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("EfficientnetUnet runs on the HIP engine only: move the module and the input to the GPU "
                           "(there is no CPU fallback; the CPU restatement lives under oracle/ for tests)")
    _lib.lib()
    if x.dtype != torch.float32:
        raise TypeError("the parity path computes in fp32; got " + str(x.dtype))
    if module._flat_params.device != x.device:
        raise RuntimeError("module and input are on different devices")
    x = x.contiguous()
    training = module.training
    eng = _engine(module, x, training)
    noise = None
    if training:
        noise = module.drop_connect_noise
        if noise is None:
            noise = torch.rand(eng.n_noise_rows, x.shape[0], device=x.device, dtype=torch.float32)
        else:
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
            if tuple(noise.shape) != (eng.n_noise_rows, x.shape[0]):
                raise ValueError(f"drop_connect_noise must be [{eng.n_noise_rows}, {x.shape[0]}]")
        module._flat_nbt += 1  # every BatchNorm's num_batches_tracked (one fused add over the flat view)
    if training and torch.is_grad_enabled():
        anchor = module._anchor(x.device)
        return _UnetFunction.apply(x, anchor, module, eng, noise)
    out = torch.empty(eng.plan.logits_shape, dtype=torch.float32, device=x.device)
    _lib.run(eng.fwd, eng.bases(module, x, out, noise=noise), _stream(x.device))
    return out
