#!/usr/bin/env python
"""Golden vectors for episode assembly: the REFERENCE's ``annotations_to_tensor`` (boxes / points through its
PromptsProcessor.apply_boxes / apply_coords) and ``LabelAnythingDataset.collate_fn`` are run on seeded ragged inputs; inputs and
outputs go to tests/golden/collate.safetensors.    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_collate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden as MG           # noqa: E402,F401  (stub finder, reference first on sys.path)

import numpy as np                        # noqa: E402
import torch                              # noqa: E402
from safetensors.torch import save_file   # noqa: E402


def ragged(rng, n_img, cats, width, max_a):
    anns = []
    for _ in range(n_img):
        d = {}
        for cid in cats:
            m = int(rng.integers(0, max_a + 1))
            d[cid] = (rng.random((m, width)) * 400).astype(np.float64) if m else np.zeros((0, width))
        anns.append(d)
    return anns


def episode(rng, m, c, a_b, a_p, h, w, classes):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    return {
        "images": torch.randn(m + 1, 3, 16, 16, generator=g),
        "prompt_masks": (torch.rand(m, c, 8, 8, generator=g) > 0.5).float(), "flag_masks": torch.randint(0, 2, (m, c), generator=g).to(torch.uint8),
        "prompt_bboxes": torch.rand(m, c, a_b, 4, generator=g), "flag_bboxes": torch.randint(0, 2, (m, c, a_b), generator=g).to(torch.uint8),
        "prompt_points": torch.rand(m, c, a_p, 2, generator=g), "flag_points": torch.randint(0, 2, (m, c, a_p), generator=g).to(torch.uint8),
        "flag_examples": torch.randint(0, 2, (m, c), generator=g).to(torch.uint8),
        "dims": torch.tensor([[h, w]] * (m + 1)), "classes": classes, "image_ids": list(range(m + 1)),
        "ground_truths": torch.randint(0, c, (m + 1, h, w), generator=g),
    }


def main():
    from label_anything.data import utils as U
    from label_anything.data.dataset import LabelAnythingDataset
    from label_anything.data.transforms import PromptsProcessor
    rng = np.random.default_rng(7)
    out = {}
    pp = PromptsProcessor(long_side_length=1024, masks_side_length=256, custom_preprocess=True)
    sizes = [(480, 640), (333, 500), (1024, 768)]
    for kind, width, enum in (("bbox", 4, U.PromptType.BBOX), ("point", 2, U.PromptType.POINT)):
        anns = ragged(rng, 3, [5, 17, 2], width, 4)
        t, f = U.annotations_to_tensor(pp, anns, sizes, enum)
        out[f"a2t.{kind}.tensor"], out[f"a2t.{kind}.flag"] = t.contiguous(), f.contiguous()
        for i, d in enumerate(anns):
            for cid, v in d.items():
                out[f"a2t.{kind}.in.{i}.{cid}"] = torch.from_numpy(np.ascontiguousarray(v))
    eps = [episode(rng, 2, 3, 2, 3, 20, 30, [[1, 4], [4]]), episode(rng, 2, 2, 4, 1, 26, 24, [[9], [9]])]
    (data, gts), _ = LabelAnythingDataset.collate_fn(None, [(e, "coco") for e in eps])
    for k, v in data.items():
        if isinstance(v, torch.Tensor):
            out["collate.out." + k] = v.contiguous()
            out["collate.dtype." + k] = torch.tensor([{torch.float32: 0, torch.uint8: 1, torch.bool: 2, torch.int64: 3}[v.dtype]])
    out["collate.out.ground_truths"] = gts.contiguous()
    for i, e in enumerate(eps):
        for k, v in e.items():
            if isinstance(v, torch.Tensor):
                out[f"collate.in.{i}.{k}"] = v.contiguous()
    save_file({k: (v.to(torch.uint8) if v.dtype == torch.bool else v) for k, v in out.items()}, os.path.join(ROOT, "tests", "golden", "collate.safetensors"))
    print("written", len(out), "tensors")


if __name__ == "__main__":
    main()
