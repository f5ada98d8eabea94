"""Image preprocessing on the device (the step before the path).

Reference pipeline (data/transforms.py:14-46, preprocess.py:109-121,240-246): PIL image -> ``torchvision...resize`` (for a
PIL image that is ``Image.resize(size, BILINEAR)``: an antialiased triangle filter evaluated in 8-bit fixed point, horizontal
pass then vertical pass) -> ``ToTensor`` -> ``(x - mean) / std`` -> zero pad to S x S (``CustomNormalize``).  Here the decoded
uint8 image goes to the GPU as it is and ``la_resample_u8`` / ``la_u8_to_chw_norm`` do the rest; the filter coefficients are
tiny per-size integer tables built on the host with the same double-precision arithmetic as PIL, so the result is bit-exact.
"""
from __future__ import annotations

import functools
import math
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

PIL_PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=256)
def pil_bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """PIL ``precompute_coeffs`` (BILINEAR, support 1) + ``normalize_coeffs_8bpc``: bounds int32 [out, 2] = (first tap,
    taps), coefficients int32 [out, ksize] with 22 fractional bits."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C cast: truncation of a value >= -0.5
    xmax = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    cnt = xmax - xmin
    j = np.arange(ksize, dtype=np.float64)[None, :]
    a = np.abs((j + xmin[:, None] - centers[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(a < 1.0, 1.0 - a, 0.0)
    w = np.where(j < cnt[:, None], w, 0.0)
    ww = w.sum(axis=1, keepdims=True)                                          # same left-to-right order matters only in the
    ww = np.zeros_like(ww)                                                     # last ulp: accumulate like PIL does
    for c in range(ksize):
        ww[:, 0] += w[:, c]
    w = np.where(ww != 0.0, w / np.where(ww == 0.0, 1.0, ww), w)
    kk = np.where(w < 0, w * (1 << PIL_PRECISION_BITS) - 0.5, w * (1 << PIL_PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, cnt], axis=1).astype(np.int32)
    return bounds, kk


def resize_shape(h: int, w: int, side: int, custom_preprocess: bool, square: bool) -> Tuple[int, int]:
    """Target (height, width): CustomResize (longest side -> side, data/utils.py:441-449) | Resize((side, side)) | Resize(side)."""
    if custom_preprocess:
        s = side * 1.0 / max(h, w)
        return int(h * s + 0.5), int(w * s + 0.5)
    if square:
        return side, side
    return (side, int(side * w / h)) if h <= w else (int(side * h / w), side)


class DevicePreprocessor:
    """uint8 HWC RGB image (CPU or device) -> normalised fp32 (3, H', W') on ``device``, like ``preprocess.load_image``."""

    def __init__(self, side: int, custom_preprocess: bool, mean: Sequence[float], std: Sequence[float], square: bool, device="cuda"):
        self.side, self.custom, self.square = int(side), bool(custom_preprocess), bool(square)
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.device = torch.device(device)
        self._tables = {}

    def _table(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            b, k = pil_bilinear_coeffs(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(np.ascontiguousarray(k)).to(self.device))
            self._tables[key] = t
        return t

    def resize_u8(self, img: torch.Tensor, nh: int, nw: int) -> torch.Tensor:
        """PIL's two-pass resample (horizontal first, then vertical) of a uint8 [H, W, C] device tensor."""
        h, w, c = img.shape
        x = img.contiguous()
        if nw != w:
            b, k = self._table(w, nw)
            out = torch.empty(h, nw, c, device=self.device, dtype=torch.uint8)
            L.resample_u8(x, h, w, c, nw, b, k, out)
            x = out
        if nh != h:
            b, k = self._table(h, nh)
            out = torch.empty(nh, nw, c, device=self.device, dtype=torch.uint8)
            L.resample_u8(x, 1, h, nw * c, nh, b, k, out)
            x = out
        return x

    def __call__(self, img_u8: torch.Tensor) -> torch.Tensor:
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
            raise ValueError("expected a uint8 [H, W, 3] RGB image")
        img = img_u8.to(self.device, non_blocking=True)
        h, w = img.shape[:2]
        nh, nw = resize_shape(h, w, self.side, self.custom, self.square)
        x = self.resize_u8(img, nh, nw)
        sh, sw = (self.side, self.side) if self.custom else (nh, nw)
        out = torch.empty(3, sh, sw, device=self.device, dtype=torch.float32)
        L.u8_to_chw_norm(x, nh, nw, sh, sw, self.mean, self.std, out)
        return out
