"""On-GPU segmentation metrics without a host sync per step.

The reference keeps four torchmetrics objects per mode (train_segmentation.py:53-63) and feeds them
`logits.argmax(dim=1)` and the labels every step (:145-159): MulticlassConfusionMatrix(ignore_index = 0 if
masked_loss, normalize="true"), JaccardIndex ("iou"), Accuracy, F1Score (multiclass, torchmetrics defaults; binary task
when num_classes == 2).  All four are functions of ONE integer histogram hist[true][pred] over all pixels; this class
accumulates that histogram on the device with the CONFUSION stage (int64, bit-exact) and derives the numbers at
`compute()` (epoch end), following torchmetrics' published definitions (third party, not installed here: the derived
formulas are restated, the histogram itself is exact by construction).
"""
from __future__ import annotations

import torch

from . import _lib
from .plan import opdefs as D
from .plan.program import Program, TRef


class SegMetrics:
    def __init__(self, num_classes: int, ignore_index: int | None = 0, device=None):
        self.C, self.ignore_index = int(num_classes), ignore_index
        self.hist = torch.zeros(self.C * self.C, dtype=torch.int64, device=device or "cuda")

    def reset(self) -> None:
        self.hist.zero_()

    @torch.no_grad()
    def update(self, predictions: torch.Tensor, labels: torch.Tensor) -> None:
        """predictions: int64 class ids (e.g. `class_mask(logits)`), labels: int64, same shape."""
        if not predictions.is_cuda:
            raise RuntimeError("SegMetrics accumulates on the GPU (there is no CPU fallback)")
        if predictions.shape != labels.shape or predictions.dtype != torch.int64 or labels.dtype != torch.int64:
            raise ValueError("predictions and labels must be int64 tensors of the same shape")
        if self.hist.device != predictions.device:
            self.hist = self.hist.to(predictions.device)
        predictions, labels = predictions.contiguous(), labels.contiguous()
        n = predictions.numel()
        prog = Program()
        prog.add("CONFUSION", PRED=TRef(D.BASE["X"], 0, (n,), "i64"), LABELS=TRef(D.BASE["Y"], 0, (n,), "i64"),
                 HIST=TRef(D.BASE["OUT"], 0, (self.C * self.C,), "i64"), COUNT=n, C=self.C)
        bases = _lib.Bases().set("X", predictions).set("Y", labels).set("OUT", self.hist)
        _lib.run(prog.pack(), bases, torch.cuda.current_stream(predictions.device).cuda_stream)

    def compute(self) -> dict:
        """{"confusion_matrix" [C,C] rows normalised over the true class, "iou", "accuracy", "f1"} as CPU tensors."""
        return metrics_from_hist(self.hist.view(self.C, self.C).cpu(), self.ignore_index)


def metrics_from_hist(h: torch.Tensor, ignore_index: int | None) -> dict:
    """h[t][p] int64 counts over ALL pixels.  Definitions (torchmetrics):
    confusion matrix: samples whose target == ignore_index removed, rows divided by their sums (0 where empty);
    multiclass IoU: macro mean of TP / (TP + FP + FN) over the classes that occur in target or prediction;
    multiclass accuracy and F1: micro (= sum TP / N); binary task (C == 2): statistics of the positive class 1."""
    C = h.shape[0]
    hd = h.double()
    cm = hd.clone()
    if ignore_index is not None and 0 <= ignore_index < C:
        cm[ignore_index, :] = 0
    rows = cm.sum(1, keepdim=True)
    cmn = torch.where(rows > 0, cm / rows.clamp(min=1), torch.zeros_like(cm))
    tp = hd.diag()
    fp = hd.sum(0) - tp
    fn = hd.sum(1) - tp
    n = hd.sum()
    if C == 2:
        tn = hd[0, 0]
        iou = tp[1] / (tp[1] + fp[1] + fn[1]) if (tp[1] + fp[1] + fn[1]) > 0 else torch.tensor(0.0, dtype=torch.float64)
        acc = (tp[1] + tn) / n if n > 0 else torch.tensor(0.0, dtype=torch.float64)
        f1 = 2 * tp[1] / (2 * tp[1] + fp[1] + fn[1]) if (2 * tp[1] + fp[1] + fn[1]) > 0 else torch.tensor(0.0, dtype=torch.float64)
    else:
        union = tp + fp + fn
        present = union > 0
        per = torch.where(present, tp / union.clamp(min=1), torch.zeros_like(tp))
        iou = per[present].mean() if present.any() else torch.tensor(0.0, dtype=torch.float64)
        acc = tp.sum() / n if n > 0 else torch.tensor(0.0, dtype=torch.float64)
        f1 = acc
    return {"confusion_matrix": cmn.float(), "iou": iou.float(), "accuracy": acc.float(), "f1": f1.float()}
