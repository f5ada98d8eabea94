"""Training objective on the device, first link of SURVEY 8f row 1: the focal term of ``LabelAnythingLoss`` with per-batch
class weighting (reference ``loss/__init__.py:67-89``, ``loss/focal.py:17-26``, ``loss/utils.py:17-43``; the training config
``parameters/trainval/coco20i/mae_noembs.yaml:24-28`` uses exactly ``{focal: {weight: 1.0}}`` with ``class_weighting: True``),
fused with its gradient with respect to the logits (``la_focal_loss``): ``train.LamTrainer`` feeds that gradient into the backward of
the decoder (and, with ``train_encoder=True``, of the image encoder)."""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib as L


class FocalLossDevice:
    def __init__(self, gamma: float = 2.0, weight: float = 1.0, class_weighting: bool = True, ignore_index: int = -100):
        self.gamma, self.class_weighting, self.ignore_index = float(gamma), bool(class_weighting), int(ignore_index)
        # LabelAnythingLoss.logits_loss multiplies by the component weight twice (loss/__init__.py:77,86)
        self.scale = float(weight) * float(weight)

    def __call__(self, logits: torch.Tensor, target: torch.Tensor, need_grad: bool = True) -> Dict[str, torch.Tensor]:
        """logits fp32 (B, C, H, W) device, target int64 (B, H, W) -> {"loss": fp32 [1], "dlogits": like logits or None,
        "class_weights": fp32 [C]}."""
        if logits.device.type != "cuda" or target.device.type != "cuda":
            raise RuntimeError("FocalLossDevice needs device tensors (there is no CPU path)")
        if (logits.dtype != torch.float32 or target.dtype != torch.int64 or logits.shape[0] != target.shape[0]
                or logits.shape[2:] != target.shape[1:]):
            raise ValueError("expected fp32 logits (B, C, H, W) and int64 target (B, H, W)")
        dev, c = logits.device, logits.shape[1]
        loss = torch.empty(1, device=dev)
        dlog = torch.empty_like(logits) if need_grad else None
        cw = torch.empty(c, device=dev)
        scratch = torch.empty((c + 2) + 2048, device=dev, dtype=torch.int64)
        L.focal_loss(logits.contiguous(), target.contiguous(), self.gamma, self.class_weighting, self.scale, self.ignore_index, loss, dlog, cw,
                     scratch)
        # "bad_targets": labels outside [0, C) that are not ignore_index (torch raises on those; here they are counted on the device and
        # contribute nothing - check it where a host sync is acceptable)
        return {"loss": loss, "dlogits": dlog, "class_weights": cw, "bad_targets": scratch[c + 1]}
