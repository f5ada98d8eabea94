"""Episode assembly on the host side of the path: ragged per-image annotations -> padded prompt tensors, and
per-episode samples -> one batch padded to a common class count / annotation count / ground-truth size.

Mirrors ``annotations_to_tensor`` (reference data/utils.py:185-245) and ``LabelAnythingDataset.collate_fn`` with its
``collate_*`` helpers (data/dataset.py:142-235, data/utils.py:272-404): same keys, shapes, padding values and dtypes (including
the reference's quirk that the collated box / point flags come out as fp32).  The mask prompts themselves are rasterised on the
device (labelanything_amd.prompts.prompt_masks_from_instances, la_prompt_masks); what is left here is list handling and padding -
no arithmetic beyond the coordinate rescale of PromptsProcessor.apply_coords / apply_boxes (data/transforms.py:174-201).
"""
from __future__ import annotations

import itertools
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .image_prep import resize_shape
from .prompts import prompt_masks_from_instances

BBOX, MASK, POINT = "bbox", "mask", "point"


def max_annotations(annotations: Sequence[Dict[Any, np.ndarray]]) -> int:
    """Largest number of annotations any (image, class) carries (data/utils.py:117-131)."""
    return max((np.asarray(v).shape[0] for img in annotations for v in img.values()), default=0)


def annotations_to_tensor(annotations: Sequence[Dict[Any, Any]], img_sizes: Sequence[Tuple[int, int]], prompt_type: str, side: int = 1024,
                          mask_side: int = 256, custom_preprocess: bool = True, device: Optional[torch.device] = None):
    """annotations[i][cat_id]: boxes float [m, 4] (x1, y1, x2, y2) / points [m, 2] (x, y) in ORIGINAL pixels, or - for masks - a
    uint8 array [k, H, W] of instance masks whose union is the prompt.  Returns (tensor, flag) shaped like the reference's:
    boxes (N, C, A, 4) + (N, C, A) uint8; points (N, C, A, 2) + (N, C, A) uint8; masks (N, C, 256, 256) fp32 + (N, C) uint8.
    Classes follow the dict order of every image (all images carry the same class ids)."""
    n, c = len(annotations), len(annotations[0])
    prompt_type = str(prompt_type).lower()
    if prompt_type == MASK:
        if device is None or torch.device(device).type != "cuda":
            raise RuntimeError("mask prompts are rasterised on the device: pass device='cuda'")
        out = torch.zeros(n, c, mask_side, mask_side, device=device)
        flag = torch.zeros(n, c, dtype=torch.uint8, device=device)
        for i, ann in enumerate(annotations):
            stacks, slots = [], []
            for m in ann.values():
                m = np.asarray(m, dtype=np.uint8)
                m = m[None] if m.ndim == 2 else m
                slots.append(list(range(len(stacks), len(stacks) + m.shape[0])))
                stacks.extend(m)
            if not stacks:
                continue
            inst = torch.from_numpy(np.stack(stacks)).to(device)
            out[i], flag[i] = prompt_masks_from_instances(inst, slots, side, mask_side, custom_preprocess)
        return out, flag
    width = 4 if prompt_type == BBOX else 2
    a = max_annotations(annotations)
    out = np.zeros((n, c, a, width), dtype=np.float32)
    flag = np.zeros((n, c, a), dtype=np.uint8)
    for i, (ann, (old_h, old_w)) in enumerate(zip(annotations, img_sizes)):
        new_h, new_w = resize_shape(old_h, old_w, side, True, False) if custom_preprocess else (side, side)
        for j, v in enumerate(ann.values()):
            v = np.asarray(v)
            if v.size == 0:
                continue
            m = v.shape[0]
            pts = v.astype(np.float64).reshape(m, -1, 2).copy()             # apply_coords works in float64 on a deep copy
            pts[..., 0] *= new_w / old_w
            pts[..., 1] *= new_h / old_h
            out[i, j, :m] = pts.reshape(m, width)
            flag[i, j, :m] = 1
    t, f = torch.from_numpy(out), torch.from_numpy(flag)
    return (t.to(device), f.to(device)) if device is not None else (t, f)


def _pad_classes(t: torch.Tensor, c_new: int, a_new: Optional[int] = None, dtype=None) -> torch.Tensor:
    shape = list(t.shape)
    shape[1] = c_new
    if a_new is not None:
        shape[2] = a_new
    out = torch.zeros(shape, dtype=t.dtype if dtype is None else dtype, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


def collate_episodes(samples: Sequence[Dict[str, Any]]) -> Tuple[Dict[str, Any], torch.Tensor]:
    """``LabelAnythingDataset.collate_fn`` without the dataset-name zip: every sample is one episode (images or embeddings
    (M+1, ...), prompt tensors (M, C_i, ...), flag_examples (M, C_i), dims (M+1, 2), classes, ground_truths (M+1, H_i, W_i)).
    Episodes are padded to the largest class count / annotation count of the batch with zero flags, ground truths to the largest
    (H, W) with -100, and flag_gts marks the classes an episode really has (background included)."""
    c_max = max(s["prompt_masks"].shape[1] for s in samples)
    dims = torch.stack([s["dims"] for s in samples])
    hmax, wmax = [int(v) for v in dims.reshape(-1, 2).max(dim=0).values.tolist()]
    gts = []
    for s in samples:
        g = s["ground_truths"]
        out = torch.full((g.shape[0], hmax, wmax), -100, dtype=torch.long)
        out[:, :g.shape[1], :g.shape[2]] = g
        gts.append(out)
    ab = max(s["prompt_bboxes"].shape[2] for s in samples)
    ap = max(s["prompt_points"].shape[2] for s in samples)
    classes = [s["classes"] for s in samples]
    flag_gts = torch.zeros(len(samples), c_max, dtype=torch.bool)
    for i, cl in enumerate(classes):
        flag_gts[i, : len(set(itertools.chain(*cl))) + 1] = True
    key = "embeddings" if "embeddings" in samples[0] else "images"
    if isinstance(samples[0][key], dict):
        images: Any = {k: torch.stack([s[key][k] for s in samples]) for k in samples[0][key]}
    else:
        images = torch.stack([s[key] for s in samples])
    data = {
        key: images,
        "prompt_points": torch.stack([_pad_classes(s["prompt_points"], c_max, ap, torch.float32) for s in samples]),
        "flag_points": torch.stack([_pad_classes(s["flag_points"], c_max, ap, torch.float32) for s in samples]),     # fp32 like the reference
        "prompt_bboxes": torch.stack([_pad_classes(s["prompt_bboxes"], c_max, ab, torch.float32) for s in samples]),
        "flag_bboxes": torch.stack([_pad_classes(s["flag_bboxes"], c_max, ab, torch.float32) for s in samples]),
        "prompt_masks": torch.stack([_pad_classes(s["prompt_masks"], c_max) for s in samples]),
        "flag_masks": torch.stack([_pad_classes(s["flag_masks"], c_max) for s in samples]),
        "flag_examples": torch.stack([_pad_classes(s["flag_examples"], c_max) for s in samples]),
        "dims": dims,
        "classes": classes,
        "intended_classes": [s["intended_classes"] for s in samples] if "intended_classes" in samples[0] else None,
        "image_ids": [s["image_ids"] for s in samples],
        "flag_gts": flag_gts,
    }
    return data, torch.stack(gts)
