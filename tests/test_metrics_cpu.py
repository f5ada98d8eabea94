"""CPU: the metric oracle against the reference's own ``to_global_multiclass`` outputs (tests/golden/metrics_remap.*, written by
tools/make_golden_metrics.py from the imported reference), the host-side lookup table and the reduction formulas."""
import json
import os

import numpy as np
import torch
from safetensors.torch import load_file

from labelanything_amd.metrics import label_lut, metrics_from_state
from oracle import metrics_oracle as MO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fixture():
    t = load_file(os.path.join(GOLD, "metrics_remap.safetensors"))
    meta = json.load(open(os.path.join(GOLD, "metrics_remap.json")))
    cats = {int(c): {} for c in meta["categories"]}
    return {k: v.numpy().astype(np.int64) for k, v in t.items()}, cats, meta["classes"]


def test_oracle_remap_matches_reference_fixture():
    t, cats, classes = _fixture()
    for compact, tag in ((True, "compact"), (False, "raw")):
        p, g = MO.to_global_multiclass(classes, cats, t["preds"], t["gt"], compact=compact)
        assert np.array_equal(p, t[f"preds_{tag}"]) and np.array_equal(g, t[f"gt_{tag}"])
    # the chained replacement is visible in the fixture: episode 2 maps local 1 -> 2 and then local 2 -> 3
    assert not (t["preds_compact"][2] == 2).any() and (t["preds"][2] == 1).any()


def test_host_lut_reproduces_the_chain():
    t, cats, classes = _fixture()
    for compact, tag in ((True, "compact"), (False, "raw")):
        for i, ci in enumerate(classes):
            lut = np.array(label_lut(ci, cats, 6, compact))
            assert np.array_equal(lut, MO.label_lut(ci, cats, 6, compact))
            for name in ("preds", "gt"):
                src = t[name][i]
                mapped = np.where((src >= 0) & (src < 6), lut[np.clip(src, 0, 5)], src)
                assert np.array_equal(mapped, t[f"{name}_{tag}"][i])


def test_reduction_formulas_match_oracle():
    rng = np.random.default_rng(5)
    for k in (2, 5, 21):
        p = rng.integers(0, k, 5000)
        g = rng.integers(0, k, 5000)
        g[rng.random(5000) < 0.1] = -100
        if k > 3:                       # leave a class entirely absent (macro weights skip it)
            p[p == 3] = 0
            g[g == 3] = 0
        cm = MO.confusion_matrix(p, g, k)
        cb = MO.binary_confusion_matrix(p, g)
        assert cm.sum() == (g != -100).sum() and cb.sum() == (g != -100).sum()
        got = metrics_from_state(torch.from_numpy(cm), torch.from_numpy(cb))
        assert abs(got["BmIoU"] - MO.jaccard_macro(cm)) < 1e-6
        assert abs(got["mIoU"] - MO.strict_mean_iou(cm)) < 1e-6
        assert abs(got["FBIoU"] - MO.binary_jaccard(cb)) < 1e-6
    # perfect prediction: every present class has IoU 1; the strict variant divides by K - 1 regardless of presence
    g = rng.integers(0, 4, 1000)
    cm = MO.confusion_matrix(g, g, 6)
    m = metrics_from_state(torch.from_numpy(cm), torch.from_numpy(MO.binary_confusion_matrix(g, g)))
    assert abs(m["BmIoU"] - 1.0) < 1e-6 and abs(m["FBIoU"] - 1.0) < 1e-6 and abs(m["mIoU"] - (6 - 1) / 5) < 1e-6


def test_out_of_range_labels_raise():
    import pytest
    with pytest.raises(RuntimeError):
        metrics_from_state(torch.zeros(3, 3, dtype=torch.int64), torch.zeros(2, 2, dtype=torch.int64), invalid=2)
    with pytest.raises(RuntimeError):
        MO.confusion_matrix(np.array([0, 5]), np.array([0, 1]), 3)


def test_iou_fixture_from_the_reference_metric_classes():
    """tests/golden/metrics_iou.* (tools/make_golden_metrics_iou.py): the REFERENCE's StrictMeanIoU / MeanIoU /
    DistributedBinaryJaccardIndex (utils/metrics.py:28-53, imported and executed) on seeded label maps over several ``update`` calls.
    Their torchmetrics base classes are a stand-in that follows the published confusion-matrix / ``_jaccard_index_reduce`` algorithm
    (torchmetrics is not installable here), so this pins the reference-specific arithmetic - the background-IoU correction, the
    ``> 0 -> 1`` clamp, ignore_index = -100, the constructor arguments of run.py:654-669 - on reference-produced numbers."""
    t = load_file(os.path.join(GOLD, "metrics_iou.safetensors"))
    meta = json.load(open(os.path.join(GOLD, "metrics_iou.json")))
    for ci, case in enumerate(meta["cases"]):
        k = case["num_classes"]
        cm = sum(MO.confusion_matrix(t[f"c{ci}_preds{u}"].numpy(), t[f"c{ci}_gt{u}"].numpy(), k) for u in range(case["updates"]))
        cb = sum(MO.binary_confusion_matrix(t[f"c{ci}_preds{u}"].numpy(), t[f"c{ci}_gt{u}"].numpy()) for u in range(case["updates"]))
        assert np.array_equal(cm, t[f"c{ci}_confmat"].numpy()) and np.array_equal(cb, t[f"c{ci}_confbin"].numpy())
        got = metrics_from_state(t[f"c{ci}_confmat"], t[f"c{ci}_confbin"])
        for name in ("mIoU", "BmIoU", "FBIoU"):
            assert abs(got[name] - case[name]) < 1e-6, (ci, name, got[name], case[name])
        assert abs(MO.strict_mean_iou(cm) - case["mIoU"]) < 1e-6 and abs(MO.binary_jaccard(cb) - case["FBIoU"]) < 1e-6
        for c in case["absent"]:
            assert cm[c].sum() == 0 and cm[:, c].sum() == 0
