"""CPU: episode assembly (labelanything_amd/collate.py) against fixtures produced by the REFERENCE's annotations_to_tensor and
LabelAnythingDataset.collate_fn (tools/make_golden_collate.py -> tests/golden/collate.safetensors)."""
import os

import numpy as np
import torch
from safetensors.torch import load_file

from labelanything_amd.collate import annotations_to_tensor, collate_episodes
from tests.helpers import GOLDEN

G = load_file(os.path.join(GOLDEN, "collate.safetensors"))
DT = {0: torch.float32, 1: torch.uint8, 2: torch.bool, 3: torch.int64}


def test_annotations_to_tensor_boxes_and_points_match_the_reference():
    sizes = [(480, 640), (333, 500), (1024, 768)]
    for kind in ("bbox", "point"):
        anns = [{cid: G[f"a2t.{kind}.in.{i}.{cid}"].numpy() for cid in (5, 17, 2)} for i in range(3)]
        t, f = annotations_to_tensor(anns, sizes, kind)
        assert t.shape == G[f"a2t.{kind}.tensor"].shape and f.dtype == torch.uint8
        assert torch.equal(f, G[f"a2t.{kind}.flag"])
        assert torch.equal(t, G[f"a2t.{kind}.tensor"])        # same float64 rescale, rounded to fp32 once: bit-exact


def test_collate_matches_the_reference():
    eps = []
    for i, classes in enumerate(([[1, 4], [4]], [[9], [9]])):
        e = {k[len(f"collate.in.{i}."):]: v for k, v in G.items() if k.startswith(f"collate.in.{i}.")}
        e["classes"], e["image_ids"] = classes, list(range(3))
        eps.append(e)
    data, gts = collate_episodes(eps)
    assert torch.equal(gts, G["collate.out.ground_truths"])
    for k, v in G.items():
        if not k.startswith("collate.out.") or k.endswith("ground_truths"):
            continue
        name = k[len("collate.out."):]
        want_dt = DT[int(G["collate.dtype." + name])]
        assert data[name].dtype == want_dt, (name, data[name].dtype, want_dt)
        assert torch.equal(data[name].to(v.dtype), v), name
    assert data["classes"] == [[[1, 4], [4]], [[9], [9]]] and data["intended_classes"] is None


def test_empty_classes_and_padding():
    anns = [{1: np.zeros((0, 4)), 2: np.array([[10.0, 20.0, 30.0, 40.0]])}]
    t, f = annotations_to_tensor(anns, [(100, 200)], "bbox", side=1024, custom_preprocess=True)
    assert t.shape == (1, 2, 1, 4) and f.tolist() == [[[0], [1]]]
    assert torch.allclose(t[0, 1, 0], torch.tensor([10.0, 20.0, 30.0, 40.0]) * 1024 / 200)
