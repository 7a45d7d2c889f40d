"""Runtime glue for the Prithvi modules (MaskedAutoencoderViT, PrithviSegmentationNet): device buffers of a planned
network + one autograd node around its two stage programs.  Mirrors engine.py (the U-Net's); PyTorch supplies device
memory, the HIP stream and the autograd edge only."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import Workspace, WorkspacePool, run_backward

_TORCH_DT = {"f32": torch.float32, "i64": torch.int64, "i32": torch.int32}


class VitEngine:
    def __init__(self, module, B: int, training: bool, mask_ratio: float, device: torch.device, want_bwd: bool | None = None,
                 want_dx: bool = False):
        plan = module._make_plan(B, training, mask_ratio, want_bwd, want_dx)
        self.plan = plan
        self.fwd = plan.fwd.pack()
        self.bwd = plan.bwd.pack() if plan.bwd is not None else None
        self.spaces = WorkspacePool(plan.ws_bytes, plan.aux_bytes, device)
        self.const = torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32, device=device)
        self.wpack = torch.zeros(plan.wpack_bytes // 4 + 65536, dtype=torch.float32, device=device)
        self.wgs = torch.empty(plan.layout.n_params, dtype=torch.float32, device=device) if plan.bwd is not None else None
        self.bwd_marks = plan.bwd_param_marks

    @property
    def resident(self) -> Workspace:
        return self.spaces.peek()

    @property
    def ws(self) -> torch.Tensor:
        return self.resident.ws

    @property
    def aux(self) -> torch.Tensor:
        return self.resident.aux

    def bases(self, module, x, out, noise, dout=None, grads=None, space: Workspace | None = None, dx=None, dout_need: int | None = None) -> _lib.Bases:
        space = space or self.resident
        b = _lib.Bases()
        b.set("WS", space.ws).set("AUX", space.aux).set("CONST", self.const).set("WPACK", self.wpack)
        b.set("PARAMS", module._flat_params).set("BUFS", module._flat_bufs)
        b.set("X", x).set("OUT", out).set("NOISE", noise)
        if self.wgs is not None:
            b.set("WGS", self.wgs)
        if dout is not None:
            plan = self.plan
            need = plan.dout_bytes if plan.dout_bytes else 4 * int(torch.Size(plan.dout_shape).numel())
            if dout_need is not None:
                need = dout_need       # the stages reading the rest of the packed buffer are skipped by the caller
            if dout.numel() * dout.element_size() < need:      # the backward program reads DOUT unchecked on the device
                raise ValueError(f"DOUT holds {dout.numel() * dout.element_size()} bytes, the backward program reads {need}")
            b.set("DOUT", dout)
        if grads is not None:
            b.set("GRADS", grads)
        if dx is not None:
            b.set("DX", dx)
        return b

    def views(self, out: torch.Tensor) -> dict:
        res = {}
        for name, t in self.plan.outputs.items():
            res[name] = out[t.off:t.off + t.nbytes].view(_TORCH_DT[t.dtype]).view(t.shape)
        return res


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _VitFunction(torch.autograd.Function):
    """Whole-network autograd node; parameter gradients go straight into the module's flat gradient buffer.
    Only the primary output (loss for the MAE, logits for the segmentation net) carries a gradient."""

    @staticmethod
    def forward(ctx, x, anchor, module, eng, noise, primary):
        out = torch.empty(eng.plan.out_bytes + 256, dtype=torch.uint8, device=x.device)
        lease = eng.spaces.lease()     # held until the backward has run (engine.py, Workspace)
        _lib.run(eng.fwd, eng.bases(module, x, out, noise, space=lease.space), _stream(x.device))
        v = eng.views(out)
        names = [primary] + [n for n in v if n != primary]
        ctx.module, ctx.eng, ctx.noise, ctx.names, ctx.out, ctx.lease = module, eng, noise, names, out, lease
        ctx.save_for_backward(x)
        outs = tuple(v[n] for n in names)
        diff = set(eng.plan.douts) if eng.plan.douts else {primary}
        ctx.mark_non_differentiable(*[o for n, o in zip(names, outs) if n not in diff])
        return outs

    @staticmethod
    def backward(ctx, dprimary, *unused):
        (x,) = ctx.saved_tensors
        dx = vit_backward_raw(ctx.module, ctx.eng, ctx.lease, ctx.noise, ctx.out, x, dict(zip(ctx.names, (dprimary,) + tuple(unused))))
        return dx, None, None, None, None, None


def vit_backward_raw(module, eng, lease, noise, out, x, gouts: dict):
    """The backward program of a VitEngine forward: `gouts` = upstream gradients by output name (None = absent).  Parameter
    gradients go into the module's flat gradient buffer; returns dX or None.  Shared by the autograd node above and the
    torch.compile custom ops (compile_ops.py)."""
    if lease.space is None:
        raise RuntimeError("backward through the same forward a second time: the saved activations have been released")
    plan = eng.plan
    primary = next(iter(gouts))
    scale = getattr(module, "_grad_scale", 1.0)
    skip, dout_need = [], None
    if plan.douts:        # several differentiable outputs: one packed buffer, absent gradients as zeros
        absent = [n for n in (plan.optional_dout_ops or {}) if gouts.get(n) is None]
        if absent and len(absent) == len(plan.optional_dout_ops) and all(t.off >= 256 for n, t in plan.douts.items() if n in absent):
            # nobody differentiates through these outputs (the reference's trainer uses the loss only): no zero-filled buffer of
            # their size, and the stages that would move it are skipped
            skip = sorted(plan.optional_dout_ops[n] for n in absent)
            dout_need = 256
        buf = torch.zeros((dout_need or plan.dout_bytes) + 256, dtype=torch.uint8, device=x.device)
        for name, g in gouts.items():
            if g is not None and name in plan.douts:
                t = plan.douts[name]
                dst = buf[t.off:t.off + t.nbytes].view(torch.float32).view(t.shape)
                dst.copy_(g.reshape(t.shape))
                if scale != 1.0:
                    dst.mul_(scale)
        dout = buf
    else:
        dout = gouts[primary].contiguous().reshape(plan.dout_shape).to(torch.float32)
        if scale != 1.0:
            dout = dout * scale
    live = module._grads_live()
    accumulate = live and not getattr(module, "_overwrite_next", False)
    module._overwrite_next = False
    grads = module._grad_buffer() if not accumulate else module._grad_scratch()
    grads.zero_()
    dx = torch.empty_like(x) if plan.want_dx else None
    bases = eng.bases(module, x, out, noise, dout=dout, grads=grads, space=lease.space, dx=dx, dout_need=dout_need)
    st = _stream(x.device)

    def run_range(a, b):        # stages [a, b) of the backward program minus the skipped ranges
        for (s0, s1) in skip:
            if a < s1 and s0 < b:
                if a < s0:
                    _lib.run(eng.bwd, bases, st, a, s0)
                a = max(a, s1)
        if a < b:
            _lib.run(eng.bwd, bases, st, a, b)

    with torch.cuda.device(x.device):
        run_backward(module, eng.bwd_marks, len(eng.bwd), run_range, grads, accumulate, lo_min=eng.plan.trainable_lo)
    lease.release()
    if accumulate:
        module._grad_buffer().add_(grads)
    if not live:
        module._publish_grads(module._no_grad_params)
    if dx is not None and scale != 1.0:
        dx.mul_(1.0 / scale)       # the data-parallel 1/world applies to parameter gradients only (engine.py)
    return dx


def vit_prepare(module, x: torch.Tensor, injected: dict, mask_ratio: float | None, trainable: bool, want_dx: bool, grad_enabled: bool):
    """(engine, noise buffer, primary output name, want_grad) for one forward: plan selection exactly as the reference's modules
    behave under torch autograd (shared by run_vit and the torch.compile custom ops)."""
    if not x.is_cuda:
        raise RuntimeError(f"{type(module).__name__} runs on the HIP engine only: move the module and the input to the GPU "
                           "(there is no CPU fallback; the CPU restatement lives under oracle/ for tests)")
    _lib.lib()
    if x.dtype != torch.float32:
        raise TypeError("the parity path computes in fp32; got " + str(x.dtype))
    if module._flat_params.device != x.device:
        raise RuntimeError("module and input are on different devices")
    B = x.shape[0]
    is_seg = mask_ratio is None
    # the segmentation head has BatchNorm / Dropout2d: its plan follows module.training (train() plans always carry the
    # backward program, eval() plans only when autograd wants one); the MAE has neither, so its plan carries a backward
    # program iff a gradient is wanted
    want_grad = grad_enabled and (trainable or want_dx)
    training = module.training if is_seg else want_grad
    want_bwd = (training or want_grad) if is_seg else want_grad
    mr = 0.0 if is_seg else float(mask_ratio)
    key = (tuple(x.shape), training, want_bwd, want_dx, mr, x.device)
    eng = module._engines.get(key)
    if eng is None:
        eng = VitEngine(module, B, training, mr, x.device, want_bwd, want_dx)
        module._engines[key] = eng
    plan = eng.plan
    noise = torch.empty(max(plan.noise_bytes // 4, 1), dtype=torch.float32, device=x.device)
    for name, t in plan.noise.items():
        n = int(np.prod(t.shape))
        dst = noise[t.off // 4:t.off // 4 + n]
        src = injected.get(name)
        if src is None:
            dst.uniform_(0.0, 1.0)     # torch.rand semantics: U[0, 1)
        else:
            if tuple(src.shape) != tuple(t.shape):
                raise ValueError(f"{name} must have shape {tuple(t.shape)}, got {tuple(src.shape)}")
            dst.copy_(src.to(device=x.device, dtype=torch.float32).reshape(-1))
    if is_seg and training:
        module._flat_nbt += 1
    return eng, noise, ("logits" if is_seg else "loss"), want_grad


def run_vit(module, x: torch.Tensor, injected: dict, mask_ratio: float | None = None) -> dict:
    """Runs the module's forward program; returns {output name: tensor}.  `injected`: noise name -> tensor or None."""
    x = x.contiguous()
    trainable = any(p.requires_grad for p in module.parameters())
    want_dx = torch.is_grad_enabled() and x.requires_grad
    eng, noise, primary, want_grad = vit_prepare(module, x, injected, mask_ratio, trainable, want_dx, torch.is_grad_enabled())
    plan = eng.plan
    if want_grad:
        anchor = module._anchor(x.device)
        outs = _VitFunction.apply(x, anchor, module, eng, noise, primary)
        names = [primary] + [n for n in plan.outputs if n != primary]
        return dict(zip(names, outs))
    out = torch.empty(plan.out_bytes + 256, dtype=torch.uint8, device=x.device)
    lease = eng.spaces.lease()
    _lib.run(eng.fwd, eng.bases(module, x, out, noise, space=lease.space), _stream(x.device))
    lease.release()
    return eng.views(out)


# ---------------------------------------------------------------------------------------------------------------------
# Separately callable methods (MaskedAutoencoderViT.forward_encoder / forward_decoder / forward_loss / random_masking):
# plans with several inputs and differentiable outputs (plan/vit_plan.py, MethodPlan).  Inputs are packed into one X
# buffer, upstream gradients into one DOUT buffer, input gradients come back in a DX buffer; parameter gradients are
# ADDED to the module's flat gradient buffer (several method nodes run in one backward pass: encoder, decoder, loss).
# ---------------------------------------------------------------------------------------------------------------------
class MethodEngine:
    def __init__(self, plan, device: torch.device):
        self.plan = plan
        self.fwd = plan.fwd.pack()
        self.bwd = plan.bwd.pack() if plan.bwd is not None else None
        self.spaces = WorkspacePool(plan.ws_bytes, plan.aux_bytes, device)
        self.const = torch.tensor(plan.const_table if plan.const_table else [0] * 8, dtype=torch.int32, device=device)
        self.wpack = torch.zeros(plan.wpack_bytes // 4 + 65536, dtype=torch.float32, device=device)
        self.wgs = torch.empty(plan.layout.n_params, dtype=torch.float32, device=device) if plan.bwd is not None else None
        self.bwd_marks = plan.bwd_param_marks

    def bases(self, module, space, xbuf, out, noise, dout=None, grads=None, dx=None) -> _lib.Bases:
        b = _lib.Bases()
        b.set("WS", space.ws).set("AUX", space.aux).set("CONST", self.const).set("WPACK", self.wpack)
        b.set("PARAMS", module._flat_params).set("BUFS", module._flat_bufs)
        b.set("X", xbuf).set("OUT", out).set("NOISE", noise)
        if self.wgs is not None:
            b.set("WGS", self.wgs)
        if dout is not None:
            b.set("DOUT", dout)
        if grads is not None:
            b.set("GRADS", grads)
        if dx is not None:
            b.set("DX", dx)
        return b


def _view(buf: torch.Tensor, t) -> torch.Tensor:
    return buf[t.off:t.off + t.nbytes].view(_TORCH_DT[t.dtype]).view(t.shape)


class _MethodFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, module, eng, noise, names, *tensors):
        plan = eng.plan
        dev = module._flat_params.device
        xbuf = torch.empty(plan.x_bytes + 256, dtype=torch.uint8, device=dev)
        for name, t in zip(names, tensors):
            ref = plan.inputs[name]
            _view(xbuf, ref).copy_(t.to(device=dev, dtype=_TORCH_DT[ref.dtype]).reshape(ref.shape))
        out = torch.empty(plan.out_bytes + 256, dtype=torch.uint8, device=dev)
        lease = eng.spaces.lease()
        _lib.run(eng.fwd, eng.bases(module, lease.space, xbuf, out, noise), _stream(dev))
        outs = tuple(_view(out, plan.outputs[n]) for n in plan.outputs)
        ctx.module, ctx.eng, ctx.noise, ctx.names, ctx.xbuf, ctx.out, ctx.lease = module, eng, noise, names, xbuf, out, lease
        ctx.mark_non_differentiable(*[o for n, o in zip(plan.outputs, outs) if n not in plan.douts])
        if eng.bwd is None:
            lease.release()
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        module, eng, lease, plan = ctx.module, ctx.eng, ctx.lease, ctx.eng.plan
        if eng.bwd is None:
            raise RuntimeError("this method was planned without a backward program")
        if lease.space is None:
            raise RuntimeError("backward through the same forward a second time: the saved activations have been released")
        dev = module._flat_params.device
        # Data-parallel reducer attached (ddp.FlatGradReducer): several method nodes (encoder, decoder, loss) run in one
        # backward pass and each ADDS to the flat gradient buffer, so no bucket is final until the last of them has run.
        # The 1/world of the mean is applied where a node's parameter gradients are added to the flat buffer (NOT to the
        # upstream gradient: along a chain of nodes it would be applied once per node, and input gradients must stay
        # d loss_rank / d input), the module is marked, and FlatGradReducer.finish() all-reduces the trainable range in one
        # collective before the optimiser reads it.
        reducing = getattr(module, "_bwd_segment_hook", None) is not None
        scale = getattr(module, "_grad_scale", 1.0) if reducing else 1.0
        if reducing:
            if getattr(module, "_bucket_reduced", False):
                raise RuntimeError("one backward mixes the fused forward with separately called methods under a data-parallel "
                                   "reducer: call ddp.finish() between them, or use one of the two paths per step")
            module._method_grads_unreduced = True
        dout = torch.zeros(plan.dout_bytes + 256, dtype=torch.uint8, device=dev)
        for name, g in zip(plan.outputs, gouts):
            if g is not None and name in plan.douts:
                ref = plan.douts[name]
                _view(dout, ref).copy_(g.to(torch.float32).reshape(ref.shape))
        dx = torch.zeros(plan.dx_bytes + 256, dtype=torch.uint8, device=dev)
        main = module._grad_buffer()
        live = module._grads_live()
        if not live or getattr(module, "_overwrite_next", False):
            main.zero_()
        module._overwrite_next = False
        scratch = module._grad_scratch()
        scratch.zero_()
        _lib.run(eng.bwd, eng.bases(module, lease.space, ctx.xbuf, ctx.out, ctx.noise, dout=dout, grads=scratch, dx=dx), _stream(dev))
        lease.release()
        main.add_(scratch, alpha=scale)
        if not live or getattr(eng, "publish_all", False):
            module._publish_grads(set() if getattr(eng, "publish_all", False) else module._no_grad_params)
        grads_in = tuple(_view(dx, plan.dins[n]).clone() if n in plan.dins else None for n in ctx.names)
        return (None, None, None, None, None) + grads_in


def run_method(module, key, make_plan, inputs: dict, injected: dict | None = None, uses_params: bool = True,
               publish_all: bool = False) -> dict:
    """Runs one separately callable method.  `key`: cache key of the plan (shapes, ratios); `make_plan(want_bwd)` builds the
    MethodPlan; `inputs`: name -> tensor (device tensors; moved / cast as the plan asks).  Returns {output name: tensor}."""
    dev = module._flat_params.device
    if dev.type != "cuda":
        raise RuntimeError(f"{type(module).__name__} runs on the HIP engine only: move the module to the GPU "
                           "(there is no CPU fallback; the CPU restatement lives under oracle/ for tests)")
    _lib.lib()
    for name, t in inputs.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name}: the HIP engine takes GPU tensors")
    needs = torch.is_grad_enabled() and (any(t.requires_grad for t in inputs.values() if t.dtype.is_floating_point)
                                         or (uses_params and any(p.requires_grad for p in module.parameters())))
    ckey = ("method",) + tuple(key) + (needs, dev)
    eng = module._engines.get(ckey)
    if eng is None:
        eng = MethodEngine(make_plan(needs), dev)
        eng.publish_all = publish_all     # every parameter of the layout may receive a gradient from this method
        module._engines[ckey] = eng
    plan = eng.plan
    noise = torch.empty(max(plan.noise_bytes // 4, 1), dtype=torch.float32, device=dev)
    for name, t in plan.noise.items():
        n = int(np.prod(t.shape))
        dst = noise[t.off // 4:t.off // 4 + n]
        src = (injected or {}).get(name)
        if src is None:
            dst.uniform_(0.0, 1.0)
        else:
            if tuple(src.shape) != tuple(t.shape):
                raise ValueError(f"{name} must have shape {tuple(t.shape)}, got {tuple(src.shape)}")
            dst.copy_(src.to(device=dev, dtype=torch.float32).reshape(-1))
    names = tuple(plan.inputs)
    for n in names:
        if tuple(inputs[n].shape) != tuple(plan.inputs[n].shape):
            raise ValueError(f"{n} must have shape {tuple(plan.inputs[n].shape)}, got {tuple(inputs[n].shape)}")
    if needs:
        outs = _MethodFunction.apply(module._anchor(dev), module, eng, noise, names, *[inputs[n] for n in names])
    else:
        with torch.no_grad():
            outs = _MethodFunction.forward(_NoCtx(), None, module, eng, noise, names, *[inputs[n] for n in names])
    return dict(zip(plan.outputs, outs))


class _NoCtx:
    """Stand-in context for a forward that needs no autograd node."""

    def mark_non_differentiable(self, *a):
        pass
