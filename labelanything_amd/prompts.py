"""Prompt tensors of an episode, built on the device (reference ``data/transforms.py:174-224``,
``data/utils.py:68-99,185-245``).

What the reference's ``PromptsProcessor`` + ``annotations_to_tensor`` + ``flags_merge`` produce per support image - the
``prompt_masks`` / ``flag_masks`` pair from the decoded instance masks, point / box coordinates rescaled to the network
input frame, and the merged ``flag_examples`` - without the per-class Python loop over CPU tensors.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .image_prep import resize_shape


def prompt_masks_from_instances(instance_masks: torch.Tensor, slots: Sequence[Sequence[int]], side: int = 1024, mask_side: int = 256,
                                custom_preprocess: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """instance_masks: uint8 [n, H, W] on the device (decoded annotations of ONE image, non-zero = object);
    slots[c] = indices of the instances of class slot c (may be empty).  Returns (prompt_masks fp32 [C, mask_side, mask_side],
    flag_masks uint8 [C]) = ``apply_masks`` of every slot + the mask branch of ``annotations_to_tensor``."""
    if instance_masks.device.type != "cuda" or instance_masks.dtype != torch.uint8 or instance_masks.dim() != 3:
        raise ValueError("instance_masks must be a uint8 [n, H, W] device tensor")
    dev = instance_masks.device
    n, h, w = instance_masks.shape
    first, count, index = [], [], []
    for s in slots:
        first.append(len(index))
        count.append(len(s))
        for i in s:
            if not 0 <= int(i) < n:
                raise IndexError(f"instance index {i} outside [0, {n})")
            index.append(int(i))
    c = len(slots)
    meta = torch.tensor(first + count + (index or [0]), dtype=torch.int32).to(dev)
    nh, nw = resize_shape(h, w, side, True, False) if custom_preprocess else (0, 0)
    out = torch.empty(c, mask_side, mask_side, device=dev, dtype=torch.float32)
    flags = torch.zeros(c, device=dev, dtype=torch.uint8)
    L.prompt_masks(instance_masks.contiguous(), meta[:c], meta[c:2 * c], meta[2 * c:], c, h, w, nh, nw, side, mask_side, out, flags)
    return out, flags


def apply_coords(coords: torch.Tensor, original_size: Tuple[int, int], side: int = 1024, custom_preprocess: bool = True) -> torch.Tensor:
    """``PromptsProcessor.torch_apply_coords`` (data/transforms.py:174-189): (x, y) in original pixels -> network input frame."""
    old_h, old_w = original_size
    new_h, new_w = resize_shape(old_h, old_w, side, True, False) if custom_preprocess else (side, side)
    out = coords.clone().float()
    out[..., 0] = out[..., 0] * (new_w / old_w)
    out[..., 1] = out[..., 1] * (new_h / old_h)
    return out


def apply_boxes(boxes: torch.Tensor, original_size: Tuple[int, int], side: int = 1024, custom_preprocess: bool = True) -> torch.Tensor:
    """``apply_boxes`` (data/transforms.py:191-201): [x1, y1, x2, y2] rows, both corners rescaled."""
    return apply_coords(boxes.reshape(-1, 2, 2), original_size, side, custom_preprocess).reshape(-1, 4)


def flags_merge(flag_masks: Optional[torch.Tensor] = None, flag_points: Optional[torch.Tensor] = None,
                flag_bboxes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``flags_merge`` (data/utils.py:68-99): an example counts for a class when any of its prompts is real; the background
    column is forced to 1."""
    if flag_masks is None and flag_points is None and flag_bboxes is None:
        raise ValueError("At least one of the flags must be provided.")
    parts: List[torch.Tensor] = []
    if flag_points is not None:
        parts.append(flag_points.any(dim=-1))
    if flag_bboxes is not None:
        parts.append(flag_bboxes.any(dim=-1))
    if flag_masks is not None:
        parts.append(flag_masks)
    out = torch.stack([p.bool() for p in parts], dim=1).any(dim=1) if len(parts) > 1 else parts[0].clone()
    out[:, 0] = 1
    return out
