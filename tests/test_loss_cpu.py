"""CPU: the focal-objective oracle against the reference's own LabelAnythingLoss outputs (tests/golden/focal_loss.safetensors,
written by tools/make_golden_loss.py from the imported reference)."""
import os

import torch
from safetensors.torch import load_file

from oracle import loss_oracle as LO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "focal_loss.safetensors")


def test_oracle_value_and_gradient_match_reference_fixture():
    t = load_file(GOLD)
    for name in ("a", "b", "c"):
        weight, cwt = float(t[f"{name}.cfg"][0]), bool(t[f"{name}.cfg"][1])
        x = t[f"{name}.logits"].clone().requires_grad_(True)
        val, cw = LO.focal_objective(x, t[f"{name}.target"], 2.0, weight, cwt)
        val.backward()
        assert abs(float(val.detach()) - float(t[f"{name}.loss"])) <= 1e-6 * max(1.0, float(t[f"{name}.loss"]))
        ref = t[f"{name}.grad"]
        assert torch.isfinite(ref).all() and torch.isfinite(x.grad).all()
        assert float((x.grad - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        if cwt:
            assert cw.shape == (x.shape[1],) and float(cw.min()) > 0
