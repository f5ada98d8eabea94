#!/usr/bin/env python
"""Golden vectors for the focal objective: the REFERENCE's LabelAnythingLoss (loss/__init__.py) is imported, run
on seeded logits / targets (with ignored pixels and -inf padded logits like Lam.postprocess_masks produces) and its value and
autograd gradient are stored in tests/golden/focal_loss.safetensors.   PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_loss.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_metrics import import_reference   # noqa: E402  (stub finder + path order)

import torch                                                # noqa: E402
from safetensors.torch import save_file                    # noqa: E402


def main():
    import_reference()
    from label_anything.loss import LabelAnythingLoss
    from oracle import loss_oracle as LO
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, (b, c, h, w, weight, cwt) in {"a": (2, 3, 24, 40, 1.0, True), "b": (1, 6, 32, 32, 0.5, True), "c": (2, 2, 16, 20, 1.0, False)}.items():
        logits = torch.randn(b, c, h, w, generator=g) * 3
        target = torch.randint(0, c, (b, h, w), generator=g)
        if name != "c":
            target[:, -4:, :] = -100                     # padded rows: ignored target, -inf logits except the background class
            logits[:, 1:, -4:, :] = float("-inf")
            logits[:, 0, -4:, :] = 0.0
            target[torch.rand(b, h, w, generator=g) < 0.05] = -100
        if name == "b":
            target[target == 4] = 0                      # a class that never occurs keeps weight 1
        x = logits.clone().requires_grad_(True)
        crit = LabelAnythingLoss({"focal": {"weight": weight}}, class_weighting=cwt)
        res = crit.logits_loss(x, target)
        val = res["value"] if "value" in res else list(res.values())[0]
        val.backward()
        ov, _ = LO.focal_objective(logits.clone().requires_grad_(True), target, 2.0, weight, cwt)
        assert abs(float(ov) - float(val)) <= 1e-6 * max(1.0, abs(float(val))), (float(ov), float(val))
        out[f"{name}.logits"], out[f"{name}.target"] = logits, target
        out[f"{name}.loss"], out[f"{name}.grad"] = val.detach().reshape(1), x.grad.clone()
        out[f"{name}.cfg"] = torch.tensor([weight, float(cwt)])
        print(name, float(val))
    save_file(out, os.path.join(ROOT, "tests", "golden", "focal_loss.safetensors"))


if __name__ == "__main__":
    main()
