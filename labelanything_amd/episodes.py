"""Synthetic few-shot episodes in the reference's batch schema.

Schema: ``BatchKeys`` (/root/reference/label_anything/data/utils.py:43-58) as produced by
``LabelAnythingDataset.collate_fn`` (data/dataset.py:219-233) and consumed by ``Lam.forward``
(models/lam.py:65-89).  Shapes follow BASELINE.md 3 / SURVEY.md 8d: images ~ N(0,1), one
axis-aligned rectangle mask per (support, class) in the fixed 256x256 prompt frame
(data/utils.py:205-206), background column forced present (data/utils.py:98).

Deterministic for a given seed (CPU torch.Generator), so tests, the golden-vector tool and
the bench all regenerate identical inputs without storing them.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

PROMPT_FRAME = 256      # prompt masks are always 256x256 regardless of image size


def make_episode(
    *,
    batch: int = 1,
    n_ways: int = 1,
    k_shots: int = 1,
    image_size: int = 1024,
    seed: int = 1234,
    prompts: Sequence[str] = ("mask",),
    n_points: int = 3,
    n_boxes: int = 2,
    embeddings_channels: Optional[int] = None,
    grid: Optional[int] = None,
    dims: Optional[Sequence[Sequence[int]]] = None,
    drop_mask_of: Optional[Sequence[int]] = None,
) -> Dict[str, torch.Tensor]:
    """Build one batch of ``batch`` N-way K-shot episodes.

    prompts: subset of {"mask", "point", "box"}.  embeddings_channels: emit precomputed
    ``embeddings`` (B, M+1, C, grid, grid) instead of ``images`` (cfg4-style bypass,
    models/lam.py:139-146).  dims: per-image original (H, W); default (S, S).
    drop_mask_of: (b, m, c) triple whose mask flag is cleared (exercises not_a_mask_embed).
    """
    g = torch.Generator().manual_seed(seed)
    b, m, c = batch, n_ways * k_shots, n_ways + 1
    s = image_size
    out: Dict[str, torch.Tensor] = {}
    if embeddings_channels is None:
        out["images"] = torch.randn(b, m + 1, 3, s, s, generator=g)
    else:
        assert grid is not None
        out["embeddings"] = torch.randn(b, m + 1, embeddings_channels, grid, grid, generator=g)

    f = PROMPT_FRAME
    masks = torch.zeros(b, m, c, f, f)
    flag_masks = torch.zeros(b, m, c, dtype=torch.uint8)
    points = torch.zeros(b, m, c, n_points, 2)
    flag_points = torch.zeros(b, m, c, n_points, dtype=torch.int64)
    boxes = torch.zeros(b, m, c, n_boxes, 4)
    flag_boxes = torch.zeros(b, m, c, n_boxes, dtype=torch.int64)
    for bi in range(b):
        for mi in range(m):
            cls = mi // k_shots + 1
            side = torch.randint(32, 193, (2,), generator=g)
            y0 = int(torch.randint(0, f - int(side[0]) + 1, (1,), generator=g))
            x0 = int(torch.randint(0, f - int(side[1]) + 1, (1,), generator=g))
            y1, x1 = y0 + int(side[0]), x0 + int(side[1])
            masks[bi, mi, cls, y0:y1, x0:x1] = 1.0
            masks[bi, mi, 0] = 1.0 - masks[bi, mi, cls]
            flag_masks[bi, mi, cls] = 1
            flag_masks[bi, mi, 0] = 1
            scale = s / f
            # points: positives inside the rectangle for the class, negatives outside for background
            for pi in range(n_points):
                u = torch.rand(2, generator=g)
                points[bi, mi, cls, pi, 0] = (x0 + u[0] * (x1 - x0)) * scale
                points[bi, mi, cls, pi, 1] = (y0 + u[1] * (y1 - y0)) * scale
                flag_points[bi, mi, cls, pi] = 1 if pi < n_points - 1 else -1
            flag_points[bi, mi, cls, n_points - 1] = -1 if n_points > 1 else 1
            # boxes: first box real (x0,y0,x1,y1), remaining padding
            boxes[bi, mi, cls, 0] = torch.tensor([x0, y0, x1, y1], dtype=torch.float32) * scale
            flag_boxes[bi, mi, cls, 0] = 1
    if drop_mask_of is not None:
        bi, mi, ci = drop_mask_of
        flag_masks[bi, mi, ci] = 0

    flags = []
    if "mask" in prompts:
        out["prompt_masks"] = masks
        out["flag_masks"] = flag_masks
        flags.append(flag_masks.bool())
    if "point" in prompts:
        out["prompt_points"] = points
        out["flag_points"] = flag_points
        flags.append((flag_points != 0).any(dim=-1))
    if "box" in prompts:
        out["prompt_bboxes"] = boxes
        out["flag_bboxes"] = flag_boxes
        flags.append((flag_boxes != 0).any(dim=-1))
    fe = torch.stack(flags, dim=0).any(dim=0)
    fe[:, :, 0] = True                      # data/utils.py:98  background always present
    out["flag_examples"] = fe.to(torch.uint8)
    if dims is None:
        out["dims"] = torch.full((b, m + 1, 2), s, dtype=torch.int64)
    else:
        d = torch.tensor(dims, dtype=torch.int64)
        out["dims"] = d.view(1, -1, 2).expand(b, m + 1, 2).contiguous() if d.dim() == 2 else d
    out["flag_gts"] = torch.ones(b, c, dtype=torch.bool)
    return out
