#!/usr/bin/env python
"""Pin of the metric row's IoU arithmetic: runs the REFERENCE's own metric classes
(/root/reference/label_anything/utils/metrics.py: ``StrictMeanIoU.compute`` :28-38, ``MeanIoU`` :41-42,
``DistributedBinaryJaccardIndex.update`` :45-53) on seeded label maps and stores inputs, confusion matrices and metric values under
tests/golden/metrics_iou.*.  Build-container tooling only.

What is the reference's and what is restated.  torchmetrics (1.7.1 in the reference's uv.lock) is not installed in this image and
cannot be installed, so the BASE classes the reference derives from are supplied here by a 70-line stand-in that follows the published
algorithm of ``torchmetrics.classification.{MulticlassJaccardIndex, BinaryJaccardIndex}``:
  * state ``confmat`` [K, K] int64, rows = target, columns = prediction; ``update`` drops ``target == ignore_index`` and adds
    ``bincount(target * K + preds, minlength=K * K)``  (``_multiclass_confusion_matrix_format / _update``);
  * ``compute`` = ``_jaccard_index_reduce(confmat, average)``: per-class ``diag / (rowsum + colsum - diag)`` with 0 / 0 -> 0 and, for
    "macro", weight 0 for classes with ``rowsum + colsum == 0``; for "binary" the [1, 1] entry over everything but [0, 0].
Everything ABOVE those base classes - the background-IoU correction of ``StrictMeanIoU``, the ``> 0 -> 1`` clamp of
``DistributedBinaryJaccardIndex``, the constructor arguments of experiment/run.py:654-669 (num_classes = K + 1, ignore_index = -100,
default average) - is the reference's own code, imported and executed.  The fixture therefore pins the reference-specific arithmetic;
the stand-in's part stays a restatement (stated as such in oracle/metrics_oracle.py and README).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_metrics_iou.py
"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402


def install_torchmetrics_standin() -> None:
    """A ``torchmetrics`` package with exactly the names utils/metrics.py imports (:5-12)."""

    def _jaccard_index_reduce(confmat, average, ignore_index=None, zero_division=0.0):
        confmat = confmat.to(torch.float32)
        if average == "binary":
            return confmat[1, 1] / (confmat[0, 1] + confmat[1, 0] + confmat[1, 1]) if float(confmat[0, 1] + confmat[1, 0] + confmat[1, 1]) else torch.tensor(zero_division)
        num = confmat.diag()
        denom = confmat.sum(0) + confmat.sum(1) - num
        iou = torch.where(denom == 0, torch.full_like(num, zero_division), num / torch.where(denom == 0, torch.ones_like(denom), denom))
        if average in (None, "none"):
            return iou
        weights = torch.ones_like(iou) if average == "macro" else confmat.sum(1)
        weights = weights.clone()
        weights[confmat.sum(1) + confmat.sum(0) == 0] = 0.0
        if ignore_index is not None and 0 <= ignore_index < confmat.shape[0]:
            weights[ignore_index] = 0.0
        return ((weights * iou) / weights.sum()).sum()

    class Metric(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default)

        def forward(self, *a, **kw):
            return self.update(*a, **kw)

    class MulticlassJaccardIndex(Metric):
        def __init__(self, num_classes, average="macro", ignore_index=None, validate_args=True, zero_division=0.0, **kw):
            super().__init__()
            self.num_classes, self.average, self.ignore_index, self.zero_division = num_classes, average, ignore_index, zero_division
            self.add_state("confmat", torch.zeros(num_classes, num_classes, dtype=torch.long), "sum")

        def update(self, preds, target):
            preds, target = preds.reshape(-1), target.reshape(-1)
            if self.ignore_index is not None:
                keep = target != self.ignore_index
                preds, target = preds[keep], target[keep]
            if bool(((preds < 0) | (preds >= self.num_classes) | (target < 0) | (target >= self.num_classes)).any()):
                raise RuntimeError("Detected more unique values than expected")
            k = self.num_classes
            self.confmat = self.confmat + torch.bincount(target.long() * k + preds.long(), minlength=k * k).reshape(k, k)

        def compute(self):
            return _jaccard_index_reduce(self.confmat, average=self.average, ignore_index=self.ignore_index, zero_division=self.zero_division)

    class BinaryJaccardIndex(Metric):
        def __init__(self, threshold=0.5, ignore_index=None, validate_args=True, zero_division=0.0, **kw):
            super().__init__()
            self.ignore_index, self.zero_division = ignore_index, zero_division
            self.add_state("confmat", torch.zeros(2, 2, dtype=torch.long), "sum")

        def update(self, preds, target):
            preds, target = preds.reshape(-1), target.reshape(-1)
            if self.ignore_index is not None:
                keep = target != self.ignore_index
                preds, target = preds[keep], target[keep]
            self.confmat = self.confmat + torch.bincount(target.long() * 2 + preds.long(), minlength=4).reshape(2, 2)

        def compute(self):
            return _jaccard_index_reduce(self.confmat, average="binary", zero_division=self.zero_division)

    def binary_jaccard_index(preds, target, threshold=0.5, ignore_index=None, **kw):
        m = BinaryJaccardIndex(ignore_index=ignore_index)
        m.update(preds, target)
        return m.compute()

    tm = types.ModuleType("torchmetrics")
    tm.__path__ = []
    cl = types.ModuleType("torchmetrics.classification")
    fn = types.ModuleType("torchmetrics.functional")
    fn.__path__ = []
    fc = types.ModuleType("torchmetrics.functional.classification")
    fc.__path__ = []
    fj = types.ModuleType("torchmetrics.functional.classification.jaccard")
    tm.Metric = Metric
    tm.MetricCollection = type("MetricCollection", (dict,), {})
    cl.BinaryJaccardIndex, cl.MulticlassJaccardIndex, cl.JaccardIndex = BinaryJaccardIndex, MulticlassJaccardIndex, MulticlassJaccardIndex
    fc.binary_jaccard_index = binary_jaccard_index
    fc.multiclass_jaccard_index = None
    fj._jaccard_index_reduce = _jaccard_index_reduce
    tm.classification, tm.functional, fn.classification, fc.jaccard = cl, fn, fc, fj
    for m in (tm, cl, fn, fc, fj):
        sys.modules[m.__name__] = m


def main():
    import tools.make_golden_metrics as G
    G._STUB_ROOTS.discard("torchmetrics")
    install_torchmetrics_standin()
    G.import_reference()
    from label_anything.utils import metrics as RM           # the REFERENCE's module (first on sys.path now)
    assert RM.__file__.startswith("/root/reference/"), RM.__file__
    from oracle import metrics_oracle as MO
    from labelanything_amd.metrics import metrics_from_state
    gen = torch.Generator().manual_seed(2024)
    out, meta = {}, {"cases": []}
    for ci, (k, n_up, absent) in enumerate(((21, 3, (7, 13)), (6, 2, ()), (81, 2, (5, 40, 41, 77)), (3, 1, ()))):
        # run.py:654-669: StrictMeanIoU / MeanIoU(num_classes = K + 1 -> here k, ignore_index = -100), DistributedBinaryJaccardIndex(-100)
        strict = RM.StrictMeanIoU(num_classes=k, ignore_index=-100)
        mean = RM.MeanIoU(num_classes=k, ignore_index=-100)
        fb = RM.DistributedBinaryJaccardIndex(ignore_index=-100)
        for u in range(n_up):
            preds = torch.randint(0, k, (2, 48, 56), generator=gen)
            gt = torch.where(torch.rand(2, 48, 56, generator=gen) < 0.55, preds, torch.randint(0, k, (2, 48, 56), generator=gen))
            for c in absent:                              # classes that never occur: macro weights drop them, the strict form does not
                preds[preds == c] = 0
                gt[gt == c] = 0
            gt[torch.rand(2, 48, 56, generator=gen) < 0.08] = -100
            strict.update(preds, gt)
            mean.update(preds, gt)
            fb.update(preds, gt)
            out[f"c{ci}_preds{u}"], out[f"c{ci}_gt{u}"] = preds.to(torch.int32), gt.to(torch.int32)
        vals = {"mIoU": float(strict.compute()), "BmIoU": float(mean.compute()), "FBIoU": float(fb.compute())}
        out[f"c{ci}_confmat"], out[f"c{ci}_confbin"] = strict.confmat.clone(), fb.confmat.clone()
        # the repo's restatements must agree with the reference-driven numbers before anything is written
        cm = sum(MO.confusion_matrix(out[f"c{ci}_preds{u}"].numpy(), out[f"c{ci}_gt{u}"].numpy(), k) for u in range(n_up))
        cb = sum(MO.binary_confusion_matrix(out[f"c{ci}_preds{u}"].numpy(), out[f"c{ci}_gt{u}"].numpy()) for u in range(n_up))
        assert np.array_equal(cm, strict.confmat.numpy()) and np.array_equal(cb, fb.confmat.numpy())
        mine = metrics_from_state(strict.confmat, fb.confmat)
        for name in vals:
            assert abs(mine[name] - vals[name]) < 1e-6, (name, mine[name], vals[name])
        assert abs(MO.strict_mean_iou(cm) - vals["mIoU"]) < 1e-6 and abs(MO.jaccard_macro(cm) - vals["BmIoU"]) < 1e-6
        meta["cases"].append({"num_classes": k, "updates": n_up, "absent": list(absent), **vals})
        print(f"case {ci}: K = {k}, {n_up} updates: {vals}")
    save_file({k_: v.contiguous() for k_, v in out.items()}, os.path.join(ROOT, "tests", "golden", "metrics_iou.safetensors"))
    json.dump(meta, open(os.path.join(ROOT, "tests", "golden", "metrics_iou.json"), "w"), indent=1)
    print("reference StrictMeanIoU / MeanIoU / DistributedBinaryJaccardIndex (on the torchmetrics stand-in) == oracle == metrics_from_state; fixture written")


if __name__ == "__main__":
    main()
