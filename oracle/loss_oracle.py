"""CPU oracle (torch fp32 + autograd) of the focal objective, SURVEY 8f.1.  TEST INFRASTRUCTURE ONLY.

Restates ``get_weight_matrix_from_labels`` (``loss/utils.py:17-43``), ``FocalLoss.__call__`` (``loss/focal.py:17-26``) and the
part of ``LabelAnythingLoss.logits_loss`` that combines them (``loss/__init__.py:67-89``: the component weight is applied twice).
PINNED: tools/make_golden_loss.py runs the imported reference ``LabelAnythingLoss`` (value and autograd gradient) on seeded
inputs and commits them as tests/golden/focal_loss.safetensors; tests/test_loss_cpu.py checks this file against that fixture.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def weight_matrix_from_labels(labels: torch.Tensor, num_classes: int, ignore_index: int = -100):
    there_is_ignore = bool((labels == ignore_index).any())
    if there_is_ignore:
        wl = labels.clone() + 1
        wl[wl == ignore_index + 1] = 0
        k = num_classes + 1
    else:
        wl, k = labels, num_classes
    weights = torch.ones(k)
    classes, counts = wl.unique(return_counts=True)
    weights[classes.long()] = 1 / torch.log(1.1 + counts / counts.sum())
    if there_is_ignore:
        weights[0] = 0
        class_weights = weights[1:]
    else:
        class_weights = weights
    return weights[wl], class_weights


def focal_objective(logits: torch.Tensor, target: torch.Tensor, gamma: float = 2.0, weight: float = 1.0, class_weighting: bool = True):
    """-> (loss scalar, class_weights or None); differentiable w.r.t. ``logits``."""
    wm, cw = weight_matrix_from_labels(target, logits.shape[1]) if class_weighting else (None, None)
    ce = F.cross_entropy(logits, target, reduction="none")
    pt = torch.exp(-ce)
    fl = torch.pow(1 - pt, gamma) * ce
    if wm is not None:
        fl = torch.pow(1 - pt, gamma) * wm * ce
    return weight * (weight * fl.mean()), cw
