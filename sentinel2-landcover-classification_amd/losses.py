"""Per-pixel losses with the reference's factory surface, computed by the HIP loss kernels.

Mirrors /root/reference/src/losses.py: `LossType`, `get_loss(config)` (:24-63), `FocalLoss`
(:69-89).  `CrossEntropyLoss` stands in for the `nn.CrossEntropyLoss(weight, label_smoothing,
ignore_index)` the reference returns for LossType.CE.  DiceLoss / dice_focal raise: they crash
in the reference too (shape error) and are out of scope.
"""
from __future__ import annotations

import enum
import typing
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .plan import opdefs as D
from .plan.program import Program, TRef

Loss = typing.Callable[[torch.Tensor, torch.Tensor], torch.Tensor]
ReduceType = typing.Literal["mean", "sum"]


class LossType(str, enum.Enum):
    CE = "ce"
    FOCAL = "focal"
    DICE = "dice"
    DICE_FOCAL = "dice_focal"


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check(logits: torch.Tensor, y: torch.Tensor):
    if not logits.is_cuda or not y.is_cuda:
        raise RuntimeError("s2k losses run on the GPU only (no CPU fallback; see oracle/losses_ref.py for the CPU oracle)")
    if logits.dim() != 4 or y.dim() != 3 or logits.dtype != torch.float32 or y.dtype != torch.int64:
        raise TypeError("expected logits [B,C,H,W] float32 and labels [B,H,W] int64")
    B, C, H, W = logits.shape
    if tuple(y.shape) != (B, H, W):
        raise ValueError(f"labels {tuple(y.shape)} do not match logits {tuple(logits.shape)}")
    return B, C, H * W


def _records(B, C, HW, mode, ignore, reduce_sum, gamma, smooth, has_alpha):
    f32 = lambda base, shape, dt="f32": TRef(D.BASE[base], 0, shape, dt)  # noqa: E731
    common = dict(B=B, C=C, HW=HW, MODE=mode, IGNORE=ignore, REDUCE_SUM=int(reduce_sum), GAMMA=gamma, SMOOTH=smooth)
    alpha = f32("CONST", (C,)) if has_alpha else None
    acc = TRef(D.BASE["WS"], 0, (2,), "f64")
    loss = TRef(D.BASE["WS"], 16, (1,), "f32")
    fwd, bwd = Program("loss_fwd"), Program("loss_bwd")
    fwd.add("LOSS_FWD", LOGITS=f32("X", (B, C, HW)), LABELS=f32("Y", (B, HW), "i64"), ALPHA=alpha, LOSS=loss, ACC=acc, **common)
    bwd.add("LOSS_BWD", LOGITS=f32("X", (B, C, HW)), LABELS=f32("Y", (B, HW), "i64"), ALPHA=alpha, ACC=acc,
            GOUT=f32("DOUT", (1,)), DLOGITS=f32("OUT", (B, C, HW)), **common)
    return fwd.pack(), bwd.pack()


class _PixelLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y, alpha, mode, ignore, reduce_sum, gamma, smooth):
        B, C, HW = _check(logits, y)
        logits, y = logits.contiguous(), y.contiguous()
        fwd, bwd = _records(B, C, HW, mode, ignore, reduce_sum, gamma, smooth, alpha is not None)
        scratch = torch.empty(32, dtype=torch.uint8, device=logits.device)  # {f64 num, f64 den, f32 loss}
        bases = _lib.Bases().set("X", logits).set("Y", y).set("WS", scratch)
        if alpha is not None:
            alpha = alpha.to(device=logits.device, dtype=torch.float32).contiguous()
            bases.set("CONST", alpha)
        _lib.run(fwd, bases, _stream(logits.device))
        ctx.bwd, ctx.alpha, ctx.scratch = bwd, alpha, scratch
        ctx.save_for_backward(logits, y)
        return scratch[16:20].view(torch.float32).clone().reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, y = ctx.saved_tensors
        dlogits = torch.empty_like(logits)
        gout = gout.to(torch.float32).contiguous().reshape(1)
        bases = _lib.Bases().set("X", logits).set("Y", y).set("WS", ctx.scratch).set("DOUT", gout).set("OUT", dlogits)
        if ctx.alpha is not None:
            bases.set("CONST", ctx.alpha)
        _lib.run(ctx.bwd, bases, _stream(logits.device))
        return dlogits, None, None, None, None, None, None, None


def _device_weights(owner, w, device):
    """Class weights on the logits' device, uploaded once per (tensor version, device): a per-step host-to-device copy is a
    synchronising call (and is not allowed while a HIP graph is being captured)."""
    if w is None:
        return None
    key = (w.data_ptr(), w._version, str(device))
    cached = getattr(owner, "_wcache", None)
    if cached is None or cached[0] != key:
        cached = (key, w.detach().to(device=device, dtype=torch.float32).contiguous())
        owner._wcache = cached
    return cached[1]


@dataclass
class FocalLoss:
    """alpha[y] * (1 - pt)^gamma * ce, pt = exp(-ce); `mean` is over ALL pixels, ignored ones
    contribute 0 (reference losses.py:77-89)."""
    alpha: torch.Tensor  # (C,)
    gamma: float
    label_smoothing: float
    ignore_index: int = -100
    reduce_type: ReduceType = "mean"

    def __call__(self, y_hat: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.reduce_type not in ("mean", "sum"):
            raise ValueError(f"Invalid reduction: {self.reduce_type}.")
        if self.gamma is None:
            raise TypeError("focal_loss_gamma must be set (the reference has no default either)")
        return _PixelLoss.apply(y_hat, y, _device_weights(self, self.alpha, y_hat.device), 1, int(self.ignore_index), self.reduce_type == "sum",
                                float(self.gamma), float(self.label_smoothing))


@dataclass
class CrossEntropyLoss:
    """nn.CrossEntropyLoss(weight, label_smoothing, ignore_index), reduction 'mean' over the
    non-ignored pixels (weighted)."""
    weight: torch.Tensor | None = None
    label_smoothing: float = 0.0
    ignore_index: int = -100

    def __call__(self, y_hat: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return _PixelLoss.apply(y_hat, y, _device_weights(self, self.weight, y_hat.device), 0, int(self.ignore_index), False, 0.0,
                                float(self.label_smoothing))


def class_mask(logits: torch.Tensor) -> torch.Tensor:
    """`logits.argmax(dim=1)` (train_segmentation.py:145) — int64 [B,H,W], first maximum wins."""
    if not logits.is_cuda or logits.dim() != 4 or logits.dtype != torch.float32:
        raise RuntimeError("class_mask expects float32 GPU logits [B,C,H,W]")
    B, C, H, W = logits.shape
    logits = logits.contiguous()
    mask = torch.empty((B, H, W), dtype=torch.int64, device=logits.device)
    p = Program("argmax")
    p.add("ARGMAX", LOGITS=TRef(D.BASE["X"], 0, (B, C, H * W)), MASK=TRef(D.BASE["OUT"], 0, (B, H * W), "i64"), B=B, C=C, HW=H * W)
    _lib.run(p.pack(), _lib.Bases().set("X", logits).set("OUT", mask), _stream(logits.device))
    return mask


def loss_class_weights(config) -> torch.Tensor | None:
    """losses.py:25-33."""
    if not config.train.weighted_loss:
        return None
    w = torch.tensor(config.train.class_distribution)
    skip_first = int(config.train.masked_loss)
    w[skip_first:] = 1 - w[skip_first:]
    assert len(w) == config.num_classes, f"{len(w)}!={config.num_classes}"
    return w


def get_loss(config) -> Loss:
    class_weights = loss_class_weights(config)
    ignore = 0 if config.train.masked_loss else -100
    lt = config.train.loss_type
    if lt == LossType.CE:
        return CrossEntropyLoss(weight=class_weights, label_smoothing=config.train.label_smoothing, ignore_index=ignore)
    if lt == LossType.FOCAL:
        return FocalLoss(alpha=class_weights if class_weights is not None else torch.tensor([1.0] * config.num_classes),
                         gamma=config.train.focal_loss_gamma, label_smoothing=config.train.label_smoothing,
                         ignore_index=ignore)
    if lt in (LossType.DICE, LossType.DICE_FOCAL):
        raise NotImplementedError("dice / dice_focal raise a shape error in the reference (losses.py:101-103); not on the hot path")
    raise ValueError(f"Unknown loss type: {lt}.\nValid options: {list(LossType)}.")
