"""GPU: la_confmat_update / SegmentationMeter against the numpy oracle (bit-exact integer counts), the reference fixture of
to_global_multiclass, edge cases, and size-independent properties at the benchmark's full resolution."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from labelanything_amd import _lib as L
from labelanything_amd.metrics import SegmentationMeter
from oracle import metrics_oracle as MO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _state(m):
    k = m.k
    return m.confmat.cpu().numpy().reshape(k, k), m.confbin.cpu().numpy().reshape(2, 2), int(m.counters.cpu()[0])


@pytest.mark.parametrize("k,shape", [(2, (1, 7, 5)), (6, (3, 33, 47)), (21, (2, 64, 64)), (81, (2, 50, 51)), (200, (1, 40, 40))])
def test_confusion_matrix_matches_oracle_bit_exact(k, shape):
    rng = np.random.default_rng(k)
    p = rng.integers(0, k, shape)
    g = rng.integers(0, k, shape)
    g[rng.random(shape) < 0.15] = -100
    m = SegmentationMeter(k)
    m.update(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda())
    m.update(torch.from_numpy(g.clip(0)).cuda(), torch.from_numpy(g).cuda())          # accumulation over two batches
    cm, cb, bad = _state(m)
    assert bad == 0
    assert np.array_equal(cm, MO.confusion_matrix(p, g, k) + MO.confusion_matrix(g.clip(0), g, k))
    assert np.array_equal(cb, MO.binary_confusion_matrix(p, g) + MO.binary_confusion_matrix(g.clip(0), g))
    got = m.compute()
    assert abs(got["mIoU"] - MO.strict_mean_iou(cm)) < 1e-6 and abs(got["BmIoU"] - MO.jaccard_macro(cm)) < 1e-6
    assert abs(got["FBIoU"] - MO.binary_jaccard(cb)) < 1e-6


def test_piecewise_constant_maps_take_the_wave_uniform_path():
    """Real label maps are blocks of equal labels: whole waves see one (target, prediction) pair."""
    k = 5
    p = np.zeros((2, 128, 256), dtype=np.int64)
    g = np.zeros_like(p)
    p[:, 32:96, 64:200] = 3
    g[:, 40:100, 50:190] = 3
    g[1, :8] = -100
    p[0, 100:, :17] = 4
    m = SegmentationMeter(k)
    m.update(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda())
    cm, cb, bad = _state(m)
    assert bad == 0 and np.array_equal(cm, MO.confusion_matrix(p, g, k)) and np.array_equal(cb, MO.binary_confusion_matrix(p, g))


def test_remap_matches_reference_fixture():
    """classes/categories -> device lookup table -> counts equal the oracle's on the REFERENCE's remapped label maps."""
    t = load_file(os.path.join(GOLD, "metrics_remap.safetensors"))
    meta = json.load(open(os.path.join(GOLD, "metrics_remap.json")))
    cats = {int(c): {} for c in meta["categories"]}
    k = len(cats) + 1
    m = SegmentationMeter(k)
    m.update(t["preds"].long().cuda(), t["gt"].long().cuda(), classes=meta["classes"], categories=cats)
    cm, cb, bad = _state(m)
    rp, rg = t["preds_compact"].numpy().astype(np.int64), t["gt_compact"].numpy().astype(np.int64)
    assert bad == 0 and np.array_equal(cm, MO.confusion_matrix(rp, rg, k)) and np.array_equal(cb, MO.binary_confusion_matrix(rp, rg))
    raw = SegmentationMeter(40)
    raw.update(t["preds"].long().cuda(), t["gt"].long().cuda(), classes=meta["classes"], categories=cats, compact=False)
    cm2, _, bad2 = _state(raw)
    assert bad2 == 0 and np.array_equal(cm2, MO.confusion_matrix(t["preds_raw"].numpy().astype(np.int64), t["gt_raw"].numpy().astype(np.int64), 40))


def test_edge_cases_ignored_everything_odd_sizes_and_bad_labels():
    m = SegmentationMeter(3)
    g = torch.full((2, 3, 3), -100, dtype=torch.int64).cuda()                 # everything ignored: nothing counted
    m.update(torch.zeros_like(g), g)
    cm, cb, bad = _state(m)
    assert cm.sum() == 0 and cb.sum() == 0 and bad == 0
    p = torch.tensor([[[0, 1, 2, 7, -3]]]).cuda()                             # 5 pixels (odd, unaligned tail), two bad labels
    g = torch.tensor([[[0, 1, 1, 1, 2]]]).cuda()
    m.update(p, g)
    cm, cb, bad = _state(m)
    assert cm.sum() == 3 and bad == 3        # 7 is outside [0,3) once (multiclass); -3 is outside both matrices
    with pytest.raises(RuntimeError, match="outside"):
        m.compute()
    m.reset()
    with pytest.raises(RuntimeError, match="device tensors"):
        m.update(torch.zeros(1, 2, 2, dtype=torch.int64), torch.zeros(1, 2, 2, dtype=torch.int64))
    with pytest.raises(ValueError):
        L.confmat_update(torch.zeros(1, 4, dtype=torch.int32).cuda(), torch.zeros(1, 4, dtype=torch.int64).cuda(), None, 3, -100,
                         m.confmat, m.confbin, m.counters)


def test_full_resolution_properties_and_fused_argmax_feed():
    """BASELINE cfg2 output size (8 x 1024 x 1024): size-independent invariants instead of an oracle pass per pixel set -
    total count = non-ignored pixels, row sums = target histogram, column sums = prediction histogram - and the meter
    consumes the int64 argmax that la_post_final writes."""
    g = torch.Generator().manual_seed(3)
    b, s, k = 8, 1024, 81
    gt = torch.randint(0, k, (b, s // 16, s // 16), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    gt[:, :5] = -100
    pred = torch.roll(gt.clamp(min=0), shifts=7, dims=2).contiguous()
    m = SegmentationMeter(k)
    m.update(pred.cuda(), gt.cuda())
    cm, cb, bad = _state(m)
    keep = gt != -100
    assert bad == 0 and cm.sum() == int(keep.sum()) == cb.sum()
    assert np.array_equal(cm.sum(1), np.bincount(gt[keep].numpy(), minlength=k))
    assert np.array_equal(cm.sum(0), np.bincount(pred[keep].numpy(), minlength=k))
    assert np.array_equal(cm, MO.confusion_matrix(pred.numpy(), gt.numpy(), k))


def test_meter_reproduces_the_reference_metric_classes():
    """SegmentationMeter (la_confmat_update + metrics_from_state) on the label maps of tests/golden/metrics_iou.* must give the confusion
    matrices and the mIoU / BmIoU / FBIoU values the REFERENCE's metric classes produced (tools/make_golden_metrics_iou.py)."""
    import json
    from safetensors.torch import load_file
    t = load_file(os.path.join(GOLD, "metrics_iou.safetensors"))
    meta = json.load(open(os.path.join(GOLD, "metrics_iou.json")))
    for ci, case in enumerate(meta["cases"]):
        m = SegmentationMeter(case["num_classes"])
        for u in range(case["updates"]):
            m.update(t[f"c{ci}_preds{u}"].long().cuda(), t[f"c{ci}_gt{u}"].long().cuda())
        k = case["num_classes"]
        assert torch.equal(m.confmat.view(k, k).cpu(), t[f"c{ci}_confmat"]) and torch.equal(m.confbin.view(2, 2).cpu(), t[f"c{ci}_confbin"])
        got = m.compute()
        for name in ("mIoU", "BmIoU", "FBIoU"):
            assert abs(got[name] - case[name]) < 1e-6, (ci, name, got[name], case[name])
