"""`torch.compile`-tolerant boundary: the whole-network forward / backward as two opaque `torch.library.custom_op`s.

The reference wraps its network in `torch.compile(model=self.net, mode="max-autotune", fullgraph=...)` unless `--type debug`
(/root/reference/src/train_segmentation.py:70-75, train_mae_prithvi.py:59-64).  Dynamo cannot trace through the ctypes launches
and the planning code of the s2k engine - and it must not: the network IS one native program.  When a module's `forward` runs
under `torch.compiler.is_compiling()`, it therefore calls `torch.ops.s2lc.unet_fwd` instead of the eager path: one graph node
with a fake (meta) implementation, whose eager implementation plans / launches exactly as `engine.run_unet` does.  Its autograd
formula is a second custom op (`s2lc::unet_bwd`), so AOTAutograd's backward graph is one node too.

What the ops hide from the compiler, on purpose:
  * the module (parameters, engines, gradient buffers) travels as an integer handle into a weak registry - the flat parameter
    buffer is read and the flat gradient buffer / the parameters' `.grad` are written inside the ops, as in the eager path;
  * BatchNorm running statistics and `num_batches_tracked` are updated inside the forward op like the parameters' gradients
    inside the backward op (torch.library accepts autograd formulas for functional schemas only, so they cannot be declared as
    mutated arguments; nothing else in the traced region reads them);
  * a leaf `anchor` tensor (requires_grad) is the input through which autograd reaches the backward op.
"""
from __future__ import annotations

import weakref

import torch

_MODULES: "weakref.WeakValueDictionary[int, torch.nn.Module]" = weakref.WeakValueDictionary()
_STATE: dict = {}          # (handle, data_ptr of the forward's output) -> (engine, lease, noise): consumed by the backward op
MAX_PENDING = 4            # forwards of ONE module whose backward has not run yet (each pins a whole activation arena: 12.5 GB at b5 bs 32)


def _park(handle: int, key, state) -> None:
    """Keep a forward's workspace lease for its backward op.  Forwards whose backward never runs (inference under grad mode) are
    bounded PER MODULE: beyond MAX_PENDING the module's own oldest pending forward is released - never another module's, whose
    backward may still come (a global cap of 64 pinned up to 64 arenas and could evict a live forward of another module)."""
    _STATE[key] = state
    mine = [k for k in _STATE if k[0] == handle]
    while len(mine) > MAX_PENDING:
        old = _STATE.pop(mine.pop(0))
        old[1].release()


def register(module) -> int:
    h = id(module)
    _MODULES[h] = module
    return h


def _module(h: int):
    m = _MODULES.get(h)
    if m is None:
        raise RuntimeError("s2lc custom op: the module behind this handle no longer exists")
    return m


@torch.library.custom_op("s2lc::unet_fwd", mutates_args=())
def unet_fwd(x: torch.Tensor, anchor: torch.Tensor, handle: int, ncls: int, want_bwd: bool, want_dx: bool) -> torch.Tensor:
    from . import _lib
    from .engine import _engine, _stream

    module = _module(handle)
    if not x.is_cuda:
        raise RuntimeError("EfficientnetUnet runs on the HIP engine only: move the module and the input to the GPU")
    if x.dtype != torch.float32:
        raise TypeError("the parity path computes in fp32; got " + str(x.dtype))
    x = x.contiguous()
    training = module.training
    eng = _engine(module, x, training, training or want_bwd, want_dx)
    noise = None
    if training:
        noise = module.drop_connect_noise
        if noise is None:
            noise = torch.rand(eng.n_noise_rows, x.shape[0], device=x.device, dtype=torch.float32)
        else:
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        module._flat_nbt += 1
    out = torch.empty(eng.plan.logits_shape, dtype=torch.float32, device=x.device)
    lease = eng.spaces.lease()
    _lib.run(eng.fwd, eng.bases(module, x, out, noise=noise, space=lease.space), _stream(x.device))
    if want_bwd:
        _park(handle, (handle, out.data_ptr()), (eng, lease, noise))
    else:
        lease.release()
    return out


@unet_fwd.register_fake
def _(x, anchor, handle, ncls, want_bwd, want_dx):
    return x.new_empty((x.shape[0], ncls, x.shape[2], x.shape[3]))


# Returns (dX or an empty tensor, the anchor's "gradient": a zero scalar).  The second output is what keeps the node alive: the
# compiler treats the op as functional, and a backward graph in which nobody uses its outputs would be dead code.
@torch.library.custom_op("s2lc::unet_bwd", mutates_args=())
def unet_bwd(dout: torch.Tensor, x: torch.Tensor, out: torch.Tensor, handle: int, want_dx: bool) -> tuple[torch.Tensor, torch.Tensor]:
    from . import _lib
    from .engine import _stream, run_backward

    module = _module(handle)
    st = _STATE.pop((handle, out.data_ptr()), None)
    if st is None:
        raise RuntimeError("s2lc::unet_bwd: no saved forward for this output (backward through the same forward a second time, or "
                           "the compiled graph copied the forward's output)")
    eng, lease, noise = st
    dout = dout.contiguous()
    scale = getattr(module, "_grad_scale", 1.0)
    if scale != 1.0:
        dout = dout * scale
    live = module._grads_live()
    accumulate = live and not getattr(module, "_overwrite_next", False)
    module._overwrite_next = False
    grads = module._grad_buffer() if not accumulate else module._grad_scratch()
    grads.zero_()
    dx = torch.empty_like(x) if want_dx else x.new_empty((0,))
    bases = eng.bases(module, x.contiguous(), None, dout=dout, noise=noise, grads=grads, space=lease.space, dx=dx if want_dx else None)
    with torch.cuda.device(x.device):
        run_backward(module, eng.bwd_marks, len(eng.bwd), lambda a, b: _lib.run(eng.bwd, bases, _stream(x.device), a, b), grads, accumulate)
    lease.release()
    if accumulate:
        module._grad_buffer().add_(grads)
    if not live:
        module._publish_grads(module._no_grad_params)
    if want_dx and scale != 1.0:
        dx.mul_(1.0 / scale)
    return dx, torch.zeros((), dtype=torch.float32, device=x.device)


@unet_bwd.register_fake
def _(dout, x, out, handle, want_dx):
    return (torch.empty_like(x) if want_dx else x.new_empty((0,))), x.new_zeros(())


def _setup(ctx, inputs, output):
    x, anchor, handle, ncls, want_bwd, want_dx = inputs
    ctx.save_for_backward(x, output)
    ctx.handle, ctx.want_dx = handle, want_dx


def _backward(ctx, g):
    x, out = ctx.saved_tensors
    dx, token = torch.ops.s2lc.unet_bwd(g, x, out, ctx.handle, ctx.want_dx)
    return (dx if ctx.want_dx else None), token, None, None, None, None


unet_fwd.register_autograd(_backward, setup_context=_setup)


def compiled_unet_forward(module, x: torch.Tensor) -> torch.Tensor:
    """What EfficientnetUnet.forward does under torch.compile: ONE opaque node (traceable by Dynamo with fullgraph=True)."""
    want_dx = x.requires_grad and torch.is_grad_enabled()
    want_bwd = torch.is_grad_enabled()
    return torch.ops.s2lc.unet_fwd(x, module._compile_anchor, module._compile_handle, module.config.num_classes, want_bwd, want_dx)


# ---------------------------------------------------------------------------------------------------------------------------------
# Prithvi: MaskedAutoencoderViT.forward -> (loss, pred, mask) and PrithviSegmentationNet.forward -> logits, the same way
# (reference: train_mae_prithvi.py:59-64 compiles `self.net` too).  Noise tensors injected for parity tests are module state and
# are read inside the op, like the engines.
# ---------------------------------------------------------------------------------------------------------------------------------
def _vit_fwd(handle: int, x: torch.Tensor, mask_ratio, trainable: bool, want_dx: bool, grad_enabled: bool):
    from . import _lib
    from .vit_engine import _stream, vit_prepare

    module = _module(handle)
    x = x.contiguous()
    injected = dict(noise=module.masking_noise)
    if mask_ratio is None:
        injected["drop_u"] = module.dropout_noise
    eng, noise, primary, want_grad = vit_prepare(module, x, injected, mask_ratio, trainable, want_dx, grad_enabled)
    out = torch.empty(eng.plan.out_bytes + 256, dtype=torch.uint8, device=x.device)
    lease = eng.spaces.lease()
    _lib.run(eng.fwd, eng.bases(module, x, out, noise, space=lease.space), _stream(x.device))
    views = {k: v.clone() for k, v in eng.views(out).items()}      # separate storages: custom-op outputs must not alias each other
    if want_grad:
        _park(handle, (handle, views[primary].data_ptr()), (eng, lease, noise, out))
    else:
        lease.release()
    return views


def _vit_bwd(handle: int, key_tensor: torch.Tensor, x: torch.Tensor, gouts: dict):
    from .vit_engine import vit_backward_raw

    module = _module(handle)
    st = _STATE.pop((handle, key_tensor.data_ptr()), None)
    if st is None:
        raise RuntimeError("s2lc backward op: no saved forward for this output")
    eng, lease, noise, out = st
    dx = vit_backward_raw(module, eng, lease, noise, out, x.contiguous(), gouts)
    return dx if dx is not None else x.new_empty((0,))


@torch.library.custom_op("s2lc::mae_fwd", mutates_args=())
def mae_fwd(x: torch.Tensor, anchor: torch.Tensor, handle: int, mask_ratio: float, trainable: bool, want_dx: bool,
            grad_enabled: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    v = _vit_fwd(handle, x, float(mask_ratio), trainable, want_dx, grad_enabled)
    return v["loss"].reshape(()), v["pred"], v["mask"]


@mae_fwd.register_fake
def _(x, anchor, handle, mask_ratio, trainable, want_dx, grad_enabled):
    s = _module(handle).spec
    return x.new_empty(()), x.new_empty((x.shape[0], s.num_patches, s.patch_dim)), x.new_empty((x.shape[0], s.num_patches))


@torch.library.custom_op("s2lc::mae_bwd", mutates_args=())
def mae_bwd(dloss: torch.Tensor, dpred: torch.Tensor, has_dpred: bool, x: torch.Tensor, loss: torch.Tensor, handle: int,
            want_dx: bool) -> tuple[torch.Tensor, torch.Tensor]:
    dx = _vit_bwd(handle, loss, x, {"loss": dloss, "pred": dpred if has_dpred else None})
    return dx, torch.zeros((), dtype=torch.float32, device=x.device)


@mae_bwd.register_fake
def _(dloss, dpred, has_dpred, x, loss, handle, want_dx):
    return (torch.empty_like(x) if want_dx else x.new_empty((0,))), x.new_zeros(())


def _mae_setup(ctx, inputs, output):
    x, anchor, handle, mask_ratio, trainable, want_dx, grad_enabled = inputs
    ctx.save_for_backward(x, output[0])
    ctx.handle, ctx.want_dx = handle, want_dx
    ctx.mark_non_differentiable(output[2])


def _mae_backward(ctx, dloss, dpred, dmask):
    x, loss = ctx.saved_tensors
    has = dpred is not None
    dx, token = torch.ops.s2lc.mae_bwd(dloss, dpred if has else dloss, has, x, loss, ctx.handle, ctx.want_dx)
    return (dx if ctx.want_dx else None), token, None, None, None, None, None


mae_fwd.register_autograd(_mae_backward, setup_context=_mae_setup)


@torch.library.custom_op("s2lc::seg_fwd", mutates_args=())
def seg_fwd(x: torch.Tensor, anchor: torch.Tensor, handle: int, ncls: int, trainable: bool, want_dx: bool, grad_enabled: bool) -> torch.Tensor:
    return _vit_fwd(handle, x, None, trainable, want_dx, grad_enabled)["logits"]


@seg_fwd.register_fake
def _(x, anchor, handle, ncls, trainable, want_dx, grad_enabled):
    return x.new_empty((x.shape[0], ncls, x.shape[3], x.shape[4]))


@torch.library.custom_op("s2lc::seg_bwd", mutates_args=())
def seg_bwd(dlogits: torch.Tensor, x: torch.Tensor, logits: torch.Tensor, handle: int, want_dx: bool) -> tuple[torch.Tensor, torch.Tensor]:
    dx = _vit_bwd(handle, logits, x, {"logits": dlogits})
    return dx, torch.zeros((), dtype=torch.float32, device=x.device)


@seg_bwd.register_fake
def _(dlogits, x, logits, handle, want_dx):
    return (torch.empty_like(x) if want_dx else x.new_empty((0,))), x.new_zeros(())


def _seg_setup(ctx, inputs, output):
    x, anchor, handle, ncls, trainable, want_dx, grad_enabled = inputs
    ctx.save_for_backward(x, output)
    ctx.handle, ctx.want_dx = handle, want_dx


def _seg_backward(ctx, g):
    x, logits = ctx.saved_tensors
    dx, token = torch.ops.s2lc.seg_bwd(g, x, logits, ctx.handle, ctx.want_dx)
    return (dx if ctx.want_dx else None), token, None, None, None, None, None


seg_fwd.register_autograd(_seg_backward, setup_context=_seg_setup)


def compiled_mae_forward(module, imgs: torch.Tensor, mask_ratio: float):
    ge = torch.is_grad_enabled()
    return torch.ops.s2lc.mae_fwd(imgs, module._compile_anchor, module._compile_handle, float(mask_ratio), module._compile_trainable,
                                  imgs.requires_grad and ge, ge)


def compiled_seg_forward(module, x: torch.Tensor) -> torch.Tensor:
    ge = torch.is_grad_enabled()
    return torch.ops.s2lc.seg_fwd(x, module._compile_anchor, module._compile_handle, module.config.num_classes, module._compile_trainable,
                                  x.requires_grad and ge, ge)
