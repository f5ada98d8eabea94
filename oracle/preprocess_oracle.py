"""CPU oracle of the image-preprocessing row.  TEST INFRASTRUCTURE ONLY (imported by tests/).

The reference resizes with ``torchvision.transforms.functional.resize`` on a PIL image (``data/transforms.py:14-25``,
``preprocess.py:109-121,240-246``), which is ``PIL.Image.resize(size[::-1], BILINEAR)`` (torchvision
``_functional_pil.resize``; torchvision itself is not installed here, Pillow - the library that does the arithmetic - is),
then ``ToTensor`` and ``CustomNormalize`` / ``Normalize`` (``data/transforms.py:28-50``).  PINNED: ``reference_preprocess``
below IS that chain executed with the real Pillow and torch ops; ``resize_numpy`` restates Pillow's 8-bit resample
(``src/libImaging/Resample.c``: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc) in numpy
and is checked against Pillow in tests/test_image_prep_cpu.py.
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

from labelanything_amd.image_prep import pil_bilinear_coeffs, resize_shape   # host-side integer tables (pure numpy)


def reference_preprocess(img_u8: np.ndarray, side: int, custom_preprocess: bool, mean, std, square: bool) -> torch.Tensor:
    """uint8 [H, W, 3] -> fp32 [3, H', W'] exactly like the reference's transform chain (Pillow + torch on the CPU)."""
    h, w = img_u8.shape[:2]
    nh, nw = resize_shape(h, w, side, custom_preprocess, square)
    img = Image.fromarray(img_u8).resize((nw, nh), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0      # ToTensor
    x = (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    if custom_preprocess:
        x = torch.nn.functional.pad(x, (0, side - nw, 0, side - nh))
    return x


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    bounds, kk = pil_bilinear_coeffs(img.shape[axis], out_size)
    x = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + x.shape[1:], dtype=np.int64)
    for o in range(out_size):
        x0, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        ss = (x[x0:x0 + cnt] * kk[o, :cnt].astype(np.int64).reshape((-1,) + (1,) * (x.ndim - 1))).sum(axis=0) + (1 << 21)
        out[o] = np.clip(ss >> 22, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_numpy(img_u8: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """Pillow's two-pass 8-bit BILINEAR resample: horizontal, then vertical."""
    h, w = img_u8.shape[:2]
    t = _resample_axis(img_u8, nw, 1) if nw != w else img_u8
    return _resample_axis(t, nh, 0) if nh != h else t


# ---- prompt masks (data/transforms.py:203-224) ------------------------------------------------------------------------------
# torchvision's resize on a TENSOR with NEAREST interpolation is torch.nn.functional.interpolate(mode="nearest") on the uint8
# tensor (torchvision/transforms/_functional_tensor.py: resize); restated with the real torch op, so this part is PINNED on
# torch itself.
def reference_apply_masks(masks, side: int = 1024, mask_side: int = 256, custom_preprocess: bool = True) -> torch.Tensor:
    """masks: list of uint8 numpy arrays [H, W] (may be empty) -> uint8 tensor [1, mask_side, mask_side] (or [ms, ms] if empty)."""
    import torch.nn.functional as F
    if len(masks) == 0:
        return torch.zeros((mask_side, mask_side), dtype=torch.uint8)
    mask = torch.as_tensor(np.logical_or.reduce(masks).astype(np.uint8)).unsqueeze(0)

    def nearest(t, size):
        return F.interpolate(t.unsqueeze(0), size=size, mode="nearest")[0]
    if custom_preprocess:
        new_h, new_w = resize_shape(masks[0].shape[0], masks[0].shape[1], side, True, False)
        mask = nearest(mask, (new_h, new_w))
        mask = F.pad(mask, (0, side - new_w, 0, side - new_h))
    return nearest(mask, (mask_side, mask_side))
