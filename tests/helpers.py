"""Test helpers shared by CPU and GPU tests."""
from __future__ import annotations

import json
import os

import torch
from safetensors.torch import load_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    t = load_file(os.path.join(GOLDEN, f"{name}.safetensors"))
    with open(os.path.join(GOLDEN, f"{name}.json")) as fh:
        meta = json.load(fh)
    return t, meta


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| over finite entries; non-finite patterns must match exactly."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), "non-finite pattern differs"
    if not fin.any():
        return 0.0
    return float((a[fin] - b[fin]).abs().max() / b[fin].abs().max().clamp_min(1e-20))


def pct_rel_err(a: torch.Tensor, b: torch.Tensor, q: float = 0.999, floor: float = 0.01) -> float:
    """q-quantile of the ELEMENT-WISE relative error |a-b| / |b| over the finite entries with |b| > floor * max|b|: the second figure next
    to the max-norm ``rel_err`` - a regression in the small-magnitude logits cannot hide under the global maximum here."""
    a = a.detach().float().cpu().flatten()
    b = b.detach().float().cpu().flatten()
    fin = torch.isfinite(b) & torch.isfinite(a)
    a, b = a[fin], b[fin]
    if b.numel() == 0:
        return 0.0
    keep = b.abs() > floor * b.abs().max()
    if not keep.any():
        return 0.0
    r = ((a[keep] - b[keep]).abs() / b[keep].abs()).double()
    if r.numel() > 4_000_000:                      # torch.quantile's input limit: an even stride keeps the distribution
        r = r[:: (r.numel() + 3_999_999) // 4_000_000]
    return float(torch.quantile(r, q))


def argmax_disagreement(logits: torch.Tensor, ref_argmax: torch.Tensor, ref_logits: torch.Tensor = None,
                        margin_rel: float = 0.0):
    """(#pixels whose argmax differs, #of those where the reference top-2 margin exceeds margin_rel*max|logit|)."""
    am = logits.argmax(dim=1).cpu()
    diff = am != ref_argmax.cpu().long()
    n_diff = int(diff.sum())
    if ref_logits is None or n_diff == 0:
        return n_diff, n_diff
    rl = ref_logits.float().cpu()
    top2 = rl.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    scale = float(rl[torch.isfinite(rl)].abs().max())
    return n_diff, int((diff & (margin > margin_rel * scale)).sum())


def reference_logits(case: dict, gold: dict, batch: dict) -> torch.Tensor:
    """Full-resolution reference logits of a golden case: the stored ones, or (full-size cases keep only the low-res logits)
    the oracle's post-processing (pinned on the reference like the rest of it) applied to the stored low-res logits."""
    if "logits" in gold:
        return gold["logits"].float()
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    return O.postprocess(geometry_for(case["cfg"]), gold["low_res_logits"].float(), batch["dims"], batch.get("flag_gts"))
