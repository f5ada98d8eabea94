"""Evaluation metrics on the device.

The reference's validation loop (experiment/run.py:686-745) takes ``preds = logits.argmax(dim=1)``, maps episode-local
labels to dataset labels (``to_global_multiclass``, data/utils.py:567-590) and feeds three torchmetrics objects
(run.py:654-669): ``StrictMeanIoU`` / ``MeanIoU`` (multiclass Jaccard over K+1 classes, ignore_index -100,
utils/metrics.py:28-43) and ``DistributedBinaryJaccardIndex`` (foreground/background IoU, :45-53).  Here the label maps stay
in HBM: ``la_confmat_update`` folds them into a (K+1)^2 int64 confusion matrix and a 2x2 one; only those few numbers are
reduced over ranks (one RCCL all-reduce) and read back by ``compute``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from .parallel import sum_over_ranks


def label_lut(classes_i: Sequence[Sequence[int]], categories: Dict[int, dict], size: int, compact: bool = True) -> List[int]:
    """``to_global_multiclass`` for one batch item as a lookup table over labels 0..size-1.

    The reference rewrites label j+1 -> dataset value for j = 0, 1, ... IN PLACE, one ``torch.where`` after the other
    (data/utils.py:583-589), so a pixel that has just been rewritten to v is rewritten again when v equals a later local
    label; walking the table through the same chain reproduces that exactly."""
    lut = list(range(size))
    cats_map = {k: i + 1 for i, k in enumerate(categories.keys())}
    longest = sorted(set(sum([list(c) for c in classes_i], [])))
    for j, v in enumerate(longest):
        value = cats_map[v] if compact else v
        lut = [value if x == j + 1 else x for x in lut]
    return lut


class SegmentationMeter:
    """Accumulates mIoU (StrictMeanIoU), BmIoU (MeanIoU) and FBIoU over batches; state lives on ``device``."""

    def __init__(self, num_classes: int, ignore_index: int = -100, device="cuda"):
        self.k = int(num_classes)                   # the reference passes len(categories) + 1 (run.py:658)
        self.ignore_index = int(ignore_index)
        self.device = torch.device(device)
        self.confmat = torch.zeros(self.k * self.k, dtype=torch.int64, device=self.device)
        self.confbin = torch.zeros(4, dtype=torch.int64, device=self.device)
        self.counters = torch.zeros(1, dtype=torch.int64, device=self.device)

    def reset(self) -> None:
        self.confmat.zero_()
        self.confbin.zero_()
        self.counters.zero_()

    def update(self, preds: torch.Tensor, gt: torch.Tensor, classes: Optional[List[List[List[int]]]] = None,
               categories: Optional[Dict[int, dict]] = None, compact: bool = True) -> None:
        """preds, gt: int64 (B, H, W) on the device (``Lam.forward_argmax(...)["argmax"]`` and the ground truth).
        With ``classes`` (the batch's ``classes`` list) and ``categories`` the labels are first mapped like
        ``to_global_multiclass(classes, categories, preds, gt)``."""
        if preds.device.type != "cuda" or gt.device.type != "cuda":
            raise RuntimeError("SegmentationMeter.update needs device tensors (there is no CPU path)")
        lut = None
        if classes is not None:
            if categories is None:
                raise ValueError("categories are required with classes")
            size = max(len(set(sum([list(c) for c in ci], []))) for ci in classes) + 1
            rows = [label_lut(ci, categories, size, compact) for ci in classes]
            lut = torch.tensor(rows, dtype=torch.int32).to(self.device, non_blocking=True)
        L.confmat_update(preds.contiguous(), gt.contiguous(), lut, self.k, self.ignore_index, self.confmat, self.confbin, self.counters)

    def all_reduce(self) -> None:
        """Sum the state over the ranks of the process group (RCCL on MI355X, gloo in the CPU tests)."""
        sum_over_ranks(self.confmat)
        sum_over_ranks(self.confbin)
        sum_over_ranks(self.counters)

    def compute(self) -> Dict[str, float]:
        state = torch.cat([self.confmat, self.confbin, self.counters]).cpu()
        return metrics_from_state(state[: self.k * self.k].view(self.k, self.k), state[self.k * self.k: self.k * self.k + 4].view(2, 2),
                                  int(state[-1]))


def metrics_from_state(confmat: torch.Tensor, confbin: torch.Tensor, invalid: int = 0) -> Dict[str, float]:
    """The reduction formulas (host, a few hundred numbers): torchmetrics 1.7.1 ``_jaccard_index_reduce`` (macro over
    the classes that occur, 0/0 -> 0) + the StrictMeanIoU correction of utils/metrics.py:31-37."""
    if invalid:
        raise RuntimeError(f"{invalid} labels outside [0, {confmat.shape[0]}) (torchmetrics raises on those)")
    k = confmat.shape[0]
    cm = confmat.to(torch.float32)
    num = cm.diag()
    denom = cm.sum(0) + cm.sum(1) - num
    iou = torch.where(denom == 0, torch.zeros_like(num), num / torch.where(denom == 0, torch.ones_like(denom), denom))
    w = torch.ones_like(iou)
    w[cm.sum(1) + cm.sum(0) == 0] = 0.0
    bmiou = ((w * iou) / w.sum()).sum()
    bg = cm[0, 0] / (cm[0, 0] + cm[0, 1:].sum() + cm[1:, 0].sum())
    miou = (bmiou * k - bg) / (k - 1)
    cb = confbin.to(torch.float32)
    d = cb[0, 1] + cb[1, 0] + cb[1, 1]
    fbiou = cb[1, 1] / d if float(d) != 0 else torch.tensor(0.0)
    return {"mIoU": float(miou), "BmIoU": float(bmiou), "FBIoU": float(fbiou)}
