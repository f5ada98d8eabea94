"""CPU restatement (numpy) of the evaluation-metric row SURVEY 8f.4.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and by
tools/make_golden_metrics.py); nothing under labelanything_amd/ or label_anything/ may import it.

What is restated and how it is pinned:
  * ``to_global_multiclass``  - reference ``label_anything/data/utils.py:567-590`` (pure torch).  PINNED: the imported
    reference function is run on seeded inputs by tools/make_golden_metrics.py and its outputs are committed under
    tests/golden/metrics_*.safetensors.
  * ``StrictMeanIoU.compute`` - reference ``label_anything/utils/metrics.py:28-38`` (the background-IoU correction) and
    ``DistributedBinaryJaccardIndex.update`` ``:45-53`` (labels > 0 -> 1).  PINNED (round 4): tools/make_golden_metrics_iou.py imports
    the reference's metric classes and runs them over several ``update`` calls; their confusion matrices and mIoU / BmIoU / FBIoU values
    are committed under tests/golden/metrics_iou.* and this module is asserted against them.
  * the confusion matrix and the macro Jaccard reduction live in **torchmetrics 1.7.1** (uv.lock:2672-2673), which is not
    installed in the build image, so that part is restated from the published algorithm
    (``torchmetrics.functional.classification.confusion_matrix._multiclass_confusion_matrix_format/_update``: drop
    ``target == ignore_index``, ``bincount(target * K + preds, minlength=K*K).reshape(K, K)``;
    ``jaccard._jaccard_index_reduce``: iou = diag / (rowsum + colsum - diag) with 0/0 -> 0, macro weights 1 except 0 for
    classes with rowsum + colsum == 0) and anchored on the reference's call sites ``experiment/run.py:448-458,654-669``
    (num_classes = K+1, average macro, ignore_index -100).  This part is a RESTATEMENT in the fixture as well: the generator supplies the
    base classes ``MulticlassJaccardIndex`` / ``BinaryJaccardIndex`` from the same published algorithm (a 70-line stand-in, stated in
    its header), so the real torchmetrics package has never been executed against these numbers - "parity unpinned" for the base
    classes only; the reference-specific arithmetic on top of them is pinned.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def to_global_multiclass(classes: List[List[List[int]]], categories: Dict[int, dict], *arrays: np.ndarray, compact: bool = True):
    """data/utils.py:567-590: episode-local label j+1 -> dataset category (compact index), applied as a CHAIN of in-place
    replacements in ascending j - a pixel rewritten to value v is rewritten again if v equals a later local label."""
    out = [np.array(a, copy=True) for a in arrays]
    cats_map = {k: i + 1 for i, k in enumerate(categories.keys())}
    for i in range(len(classes)):
        longest = sorted(set(sum(classes[i], [])))
        for j, v in enumerate(longest):
            value = cats_map[v] if compact else v
            for a in out:
                a[i] = np.where(a[i] == j + 1, value, a[i])
    return out


def label_lut(classes_i: Sequence[Sequence[int]], categories: Dict[int, dict], size: int, compact: bool = True) -> np.ndarray:
    """The same chain collapsed into one lookup table for labels 0..size-1 (what the device kernel consumes)."""
    lut = np.arange(size, dtype=np.int64)
    cats_map = {k: i + 1 for i, k in enumerate(categories.keys())}
    longest = sorted(set(sum([list(c) for c in classes_i], [])))
    for j, v in enumerate(longest):
        value = cats_map[v] if compact else v
        lut[lut == j + 1] = value
    return lut


def confusion_matrix(preds: np.ndarray, target: np.ndarray, num_classes: int, ignore_index: int = -100) -> np.ndarray:
    """torchmetrics 1.7.1 multiclass confusion matrix: rows = target, columns = prediction."""
    p, t = preds.reshape(-1).astype(np.int64), target.reshape(-1).astype(np.int64)
    keep = t != ignore_index
    p, t = p[keep], t[keep]
    if ((p < 0) | (p >= num_classes) | (t < 0) | (t >= num_classes)).any():
        raise RuntimeError("label outside [0, num_classes)")          # torchmetrics' validate_args raises too
    return np.bincount(t * num_classes + p, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def binary_confusion_matrix(preds: np.ndarray, target: np.ndarray, ignore_index: int = -100) -> np.ndarray:
    """DistributedBinaryJaccardIndex.update (metrics.py:45-53) + torchmetrics binary confusion matrix."""
    p, t = preds.reshape(-1).astype(np.int64).copy(), target.reshape(-1).astype(np.int64).copy()
    p[p > 0] = 1
    t[t > 0] = 1
    keep = t != ignore_index
    return confusion_matrix(p[keep], t[keep], 2, ignore_index)


def jaccard_macro(confmat: np.ndarray) -> float:
    """_jaccard_index_reduce(average="macro", ignore_index=-100 -> outside [0, K), zero_division=0)."""
    cm = confmat.astype(np.float32)
    num = np.diag(cm)
    denom = cm.sum(0) + cm.sum(1) - num
    iou = np.where(denom == 0, np.float32(0), num / np.where(denom == 0, np.float32(1), denom)).astype(np.float32)
    w = np.ones_like(iou)
    w[cm.sum(1) + cm.sum(0) == 0] = 0
    return float(((w * iou) / w.sum()).sum())


def strict_mean_iou(confmat: np.ndarray) -> float:
    """metrics.py:28-38: macro IoU with the background class taken out again, normalised by K - 1."""
    k = confmat.shape[0]
    cm = confmat.astype(np.float32)
    bg = cm[0, 0] / (cm[0, 0] + cm[0, 1:].sum() + cm[1:, 0].sum())
    return float((np.float32(jaccard_macro(confmat)) * k - bg) / (k - 1))


def binary_jaccard(confmat2: np.ndarray) -> float:
    cm = confmat2.astype(np.float32)
    d = cm[0, 1] + cm[1, 0] + cm[1, 1]
    return float(cm[1, 1] / d) if d != 0 else 0.0
