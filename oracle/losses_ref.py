"""TEST INFRASTRUCTURE ONLY — CPU oracle: per-pixel CE / focal loss and the class mask.

Restates, in explicit log-softmax arithmetic (no F.cross_entropy), what
  /root/reference/src/losses.py:24-63  (get_loss: weights, ignore_index = 0 if masked else -100)
  /root/reference/src/losses.py:69-89  (FocalLoss.__call__)
  torch.nn.CrossEntropyLoss(weight, label_smoothing, ignore_index)  (losses.py:36-40)
  logits.argmax(dim=1)  (/root/reference/src/train_segmentation.py:145,206)
compute.  Pinned by tests/golden/loss_cases.npz (generated from the imported reference).
"""
from __future__ import annotations

import torch


def _per_pixel_ce(logits: torch.Tensor, y: torch.Tensor, weight, label_smoothing: float, ignore_index: int):
    """F.cross_entropy(..., reduction='none') semantics.  logits [B,C,H,W], y [B,H,W] int64.

    Returns (ce [B,H,W], w_y [B,H,W], valid [B,H,W]).  With smoothing eps and class weights w:
      ce = (1-eps) * w[y] * (-logp[y]) + eps/C * sum_c w[c] * (-logp[c]);  0 where y == ignore_index.
    """
    C = logits.shape[1]
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    valid = y != ignore_index
    ys = torch.where(valid, y, torch.zeros_like(y))
    w = torch.ones(C, dtype=logits.dtype) if weight is None else weight.to(logits.dtype)
    nll = -logp.gather(1, ys.unsqueeze(1)).squeeze(1)
    w_y = w[ys]
    ce = (1.0 - label_smoothing) * w_y * nll
    if label_smoothing > 0.0:
        smooth = -(logp * w.view(1, C, 1, 1)).sum(1)
        ce = ce + (label_smoothing / C) * smooth
    zero = torch.zeros_like(ce)
    return torch.where(valid, ce, zero), torch.where(valid, w_y, zero), valid


def cross_entropy(logits, y, weight=None, label_smoothing: float = 0.0, ignore_index: int = -100):
    """nn.CrossEntropyLoss 'mean': sum(ce) / sum_{valid} w[y]  (NaN when nothing is valid, as torch)."""
    ce, w_y, _ = _per_pixel_ce(logits, y, weight, label_smoothing, ignore_index)
    return ce.sum() / w_y.sum()


def focal(logits, y, alpha: torch.Tensor, gamma: float, label_smoothing: float = 0.0,
          ignore_index: int = -100, reduce_type: str = "mean"):
    """FocalLoss.__call__ losses.py:77-89: unweighted CE per pixel, pt = exp(-ce),
    alpha[y] * (1-pt)^gamma * ce, then mean over ALL B*H*W pixels (ignored ones contribute 0).
    alpha.gather(0, y) in the reference indexes with the raw label, so ignore_index must be a
    valid class id (0) or absent from y."""
    ce, _, _ = _per_pixel_ce(logits, y, None, label_smoothing, ignore_index)
    pt = torch.exp(-ce)
    a = alpha.to(logits.dtype)[y]
    fl = a * (1.0 - pt) ** gamma * ce
    return fl.mean() if reduce_type == "mean" else fl.sum()


def class_mask(logits: torch.Tensor) -> torch.Tensor:
    """argmax over classes, first maximum wins; int64 [B,H,W]."""
    best = logits[:, 0]
    idx = torch.zeros_like(best, dtype=torch.int64)
    for c in range(1, logits.shape[1]):
        better = logits[:, c] > best
        best = torch.where(better, logits[:, c], best)
        idx = torch.where(better, torch.full_like(idx, c), idx)
    return idx


def loss_class_weights(class_distribution, masked_loss: bool) -> torch.Tensor:
    """losses.py:26-29: w = p; w[skip_first:] = 1 - p[skip_first:]."""
    w = torch.tensor(class_distribution, dtype=torch.float32)
    k = int(masked_loss)
    w[k:] = 1 - w[k:]
    return w
