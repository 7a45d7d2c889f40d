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


class Workspace:
    """The per-forward device state of a planned network: activation arena + statistics accumulators.  A training
    forward LEASES one until its backward has run (or its autograd node dies), so fwd, fwd, bwd, bwd — two micro-batches,
    two losses, a logging forward between forward and backward — never overwrite each other's saved activations."""

    def __init__(self, ws_bytes: int, aux_bytes: int, device: torch.device):
        pad = 256
        self.ws = torch.empty(ws_bytes + pad, dtype=torch.uint8, device=device)
        self.aux = torch.zeros(max(aux_bytes, 8) + pad, dtype=torch.uint8, device=device)


class WorkspaceLease:
    """Returns the workspace to its engine's pool when released or garbage-collected with the autograd node."""

    def __init__(self, pool: list, space: Workspace):
        self.pool, self.space = pool, space

    def release(self) -> None:
        if self.space is not None:
            self.pool.append(self.space)
            self.space = None

    def __del__(self):
        self.release()


class WorkspacePool:
    def __init__(self, ws_bytes: int, aux_bytes: int, device: torch.device):
        self.args = (ws_bytes, aux_bytes, device)
        self.free: list[Workspace] = [Workspace(*self.args)]
        self.allocated = 1

    def lease(self) -> WorkspaceLease:
        if not self.free:       # an earlier forward still owns the last one (its backward is outstanding)
            self.free.append(Workspace(*self.args))
            self.allocated += 1
        return WorkspaceLease(self.free, self.free.pop())

    def peek(self) -> Workspace:
        if not self.free:
            self.free.append(Workspace(*self.args))
            self.allocated += 1
        return self.free[-1]


class UnetEngine:
    """Buffers + packed programs for one (B, H, W, training) shape of one module."""

    def __init__(self, module, B: int, H: int, W: int, training: bool, device: torch.device, want_bwd: bool | None = None,
                 want_dx: bool = False):
        plan = module._make_plan(B, H, W, training, want_bwd, want_dx)
        self.want_dx = want_dx
        self.plan = plan
        self.fwd = plan.fwd.pack()
        self.bwd = plan.bwd.pack() if plan.bwd is not None else None
        self.device = device
        self.spaces = WorkspacePool(plan.ws_bytes, plan.aux_bytes, device)
        self.const = torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32, device=device)
        self.wpack = torch.zeros(plan.wpack_bytes // 4 + 65536, dtype=torch.float32, device=device)  # + slack: A-tile loads may overrun
        self.wgs = torch.empty(plan.layout.n_params, dtype=torch.float32, device=device) if plan.bwd is not None else None
        self.n_noise_rows = plan.n_noise_rows
        self.B = B
        self.bwd_marks = plan.bwd_param_marks

    @property
    def resident(self) -> Workspace:
        """The workspace a fresh forward would take (tests / profilers that drive the programs by hand)."""
        return self.spaces.peek()

    @property
    def ws(self) -> torch.Tensor:
        return self.resident.ws

    @property
    def aux(self) -> torch.Tensor:
        return self.resident.aux

    def bases(self, module, x, out, dout=None, noise=None, grads=None, space: Workspace | None = None, dx=None) -> _lib.Bases:
        space = space or self.resident
        b = _lib.Bases()
        b.set("WS", space.ws).set("AUX", space.aux).set("CONST", self.const).set("WPACK", self.wpack)
        b.set("PARAMS", module._flat_params).set("BUFS", module._flat_bufs)
        b.set("X", x).set("OUT", out)
        if self.wgs is not None:
            b.set("WGS", self.wgs)
        if dout is not None:
            need = 4 * int(torch.Size(self.plan.logits_shape).numel())
            if dout.numel() * dout.element_size() < need:      # the backward program reads DOUT unchecked on the device
                raise ValueError(f"DOUT holds {dout.numel() * dout.element_size()} bytes, the backward program reads {need}")
            b.set("DOUT", dout)
        if noise is not None:
            b.set("NOISE", noise)
        if grads is not None:
            b.set("GRADS", grads)
        if dx is not None:
            b.set("DX", dx)
        return b


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _engine(module, x: torch.Tensor, training: bool, want_bwd: bool, want_dx: bool = False) -> UnetEngine:
    key = (tuple(x.shape), training, want_bwd, want_dx, x.device)
    eng = module._engines.get(key)
    if eng is None:
        B, _, H, W = x.shape
        eng = UnetEngine(module, B, H, W, training, x.device, want_bwd, want_dx)
        module._engines[key] = eng
    return eng


class _UnetFunction(torch.autograd.Function):
    """Whole-network autograd node.  Parameter gradients are written straight into the module's
    flat gradient buffer by the backward program (never returned through autograd: 900+
    per-tensor accumulations would cost more host time than the GPU step)."""

    @staticmethod
    def forward(ctx, x, anchor, module, eng, noise):
        out = torch.empty(eng.plan.logits_shape, dtype=torch.float32, device=x.device)
        lease = eng.spaces.lease()
        bases = eng.bases(module, x, out, noise=noise, space=lease.space)
        _lib.run(eng.fwd, bases, _stream(x.device))
        ctx.module, ctx.eng, ctx.noise, ctx.lease = module, eng, noise, lease
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        module, eng, lease = ctx.module, ctx.eng, ctx.lease
        if lease.space is None:
            raise RuntimeError("backward through the same forward a second time: the saved activations have been released "
                               "(the s2k engine keeps them for one backward, like autograd without retain_graph)")
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
        dx = torch.empty_like(x) if eng.want_dx else None      # gradient w.r.t. the input (planned only when x.requires_grad)
        bases = eng.bases(module, x, None, dout=dout, noise=ctx.noise, grads=grads, space=lease.space, dx=dx)
        st = _stream(x.device)
        with torch.cuda.device(x.device):
            run_backward(module, eng.bwd_marks, len(eng.bwd), lambda a, b: _lib.run(eng.bwd, bases, st, a, b), grads, accumulate)
        lease.release()     # stream-ordered: the next forward that takes this workspace is enqueued behind this backward
        if accumulate:
            module._grad_buffer().add_(grads)
        if not live:
            module._publish_grads(module._no_grad_params)
        if dx is not None and scale != 1.0:
            # the 1/world of the data-parallel mean belongs to the PARAMETER gradients only: whatever sits in front of the
            # module (an input adapter under its own DDP, a saliency map) must see d loss_rank / d x, as with torch DDP
            dx.mul_(1.0 / scale)
        return dx, None, None, None, None


def _note_bucket_reduction(module) -> None:
    """Bookkeeping for ddp.FlatGradReducer: this backward hands its gradients to the bucket hook segment by segment.  The
    separately callable methods (vit_engine._MethodFunction) instead ADD local gradients to the flat buffer and leave the
    reduction to FlatGradReducer.finish(); both in one backward would reduce the bucketed part twice."""
    if getattr(module, "_method_grads_unreduced", False):
        raise RuntimeError("one backward mixes the fused forward with separately called methods under a data-parallel reducer: "
                           "call ddp.finish() between them, or use one of the two paths per step")
    module._bucket_reduced = True


def run_backward(module, marks, n_ops: int, run_range, grads: torch.Tensor, accumulate: bool, lo_min: int = 0) -> None:
    """Enqueue a whole-network backward program, with or without the data-parallel reducer (shared by the eager autograd nodes of
    engine.py / vit_engine.py and the torch.compile custom ops, so their checks cannot drift apart).

      no reducer, or inside `FlatGradReducer.no_sync()`   the program runs in one go; under no_sync the module is marked: its flat
                                                          buffer holds LOCAL gradients that a later backward / finish() reduces
      reducer, fresh gradients                            segment by segment, each final suffix bucket handed to the hook
      reducer, accumulating onto live gradients           (gradient accumulation: the last micro-batch of a no_sync sequence) the
                                                          program runs in one go into the scratch buffer, the caller adds it to the
                                                          flat buffer, and finish() reduces the SUM in one collective - torch DDP's
                                                          semantics: one reduction of the accumulated gradients per optimiser step
    `run_range(a, b)` enqueues stages [a, b); `marks` = (a, b, lo, hi) backward segments (plan.bwd_param_marks)."""
    hook = getattr(module, "_bwd_segment_hook", None)
    if hook is None or getattr(module, "_no_sync", False):
        if hook is not None:
            module._method_grads_unreduced = True      # local gradients: reduced as a whole by finish()
        run_range(0, n_ops)
        return
    if accumulate or getattr(module, "_method_grads_unreduced", False):
        if getattr(module, "_bucket_reduced", False):
            raise RuntimeError("a second backward accumulates onto gradients that were already all-reduced bucket by bucket in this "
                               "step: run every micro-batch but the last inside `with reducer.no_sync():` (as with torch DDP)")
        module._method_grads_unreduced = True
        run_range(0, n_ops)
        return
    _note_bucket_reduction(module)
    for (a, b, lo, hi) in marks:
        run_range(a, b)
        hook(max(lo, lo_min), hi, grads)


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
    # train() plans always carry the backward program; eval() plans only when autograd wants one (torch differentiates an
    # eval-mode module just the same: BatchNorm on its running statistics, no drop-connect)
    want_dx = torch.is_grad_enabled() and x.requires_grad
    differentiate = torch.is_grad_enabled() and (want_dx or any(p.requires_grad for p in module.parameters()))
    want_bwd = training or differentiate
    eng = _engine(module, x, training, want_bwd, want_dx)
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
    if differentiate:
        anchor = module._anchor(x.device)
        return _UnetFunction.apply(x, anchor, module, eng, noise)
    out = torch.empty(eng.plan.logits_shape, dtype=torch.float32, device=x.device)
    lease = eng.spaces.lease()
    _lib.run(eng.fwd, eng.bases(module, x, out, noise=noise, space=lease.space), _stream(x.device))
    lease.release()
    return out
