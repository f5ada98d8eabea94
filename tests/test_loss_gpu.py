"""GPU: la_focal_loss (value + gradient, fused) against the reference fixture and the torch-autograd oracle."""
import os

import pytest
import torch
from safetensors.torch import load_file

from labelanything_amd.loss import FocalLossDevice
from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "focal_loss.safetensors")


def test_focal_loss_matches_reference_fixture():
    t = load_file(GOLD)
    for name in ("a", "b", "c"):
        weight, cwt = float(t[f"{name}.cfg"][0]), bool(t[f"{name}.cfg"][1])
        out = FocalLossDevice(2.0, weight, cwt)(t[f"{name}.logits"].cuda(), t[f"{name}.target"].cuda())
        ref_l, ref_g = float(t[f"{name}.loss"]), t[f"{name}.grad"]
        assert abs(float(out["loss"]) - ref_l) <= 2e-6 * max(1.0, ref_l), name
        g = out["dlogits"].cpu()
        assert torch.isfinite(g).all()
        assert float((g - ref_g).abs().max()) <= 5e-6 * float(ref_g.abs().max()), name


@pytest.mark.parametrize("shape", [(1, 2, 1024, 1024), (3, 21, 200, 333), (2, 6, 64, 64)])
def test_focal_loss_matches_autograd_oracle(shape):
    b, c, h, w = shape
    g = torch.Generator().manual_seed(c)
    logits = torch.randn(b, c, h, w, generator=g) * 4
    target = torch.randint(0, c, (b, h, w), generator=g)
    target[torch.rand(b, h, w, generator=g) < 0.1] = -100
    if c > 3:
        target[target == 2] = 1                             # class 2 absent -> weight 1
    x = logits.clone().requires_grad_(True)
    val, cw = LO.focal_objective(x, target, 2.0, 1.0, True)
    val.backward()
    out = FocalLossDevice(2.0, 1.0, True)(logits.cuda(), target.cuda())
    assert abs(float(out["loss"]) - float(val.detach())) <= 5e-6 * max(1.0, float(val.detach()))
    assert float((out["class_weights"].cpu() - cw).abs().max()) <= 2e-6 * float(cw.max())
    assert float((out["dlogits"].cpu() - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())
    nog = FocalLossDevice(2.0, 1.0, True)(logits.cuda(), target.cuda(), need_grad=False)
    assert nog["dlogits"] is None and float(nog["loss"]) == float(out["loss"])        # deterministic reduction


def test_focal_loss_rejects_bad_inputs():
    f = FocalLossDevice()
    with pytest.raises(RuntimeError, match="device tensors"):
        f(torch.zeros(1, 2, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        f(torch.zeros(1, 2, 4, 4).cuda(), torch.zeros(1, 4, 5, dtype=torch.int64).cuda())


def test_out_of_range_targets_are_counted_not_silently_dropped():
    """torch's cross_entropy raises on labels outside [0, C); the device kernel counts them (result["bad_targets"]) and lets them
    contribute nothing - the caller checks the counter where a host sync is acceptable."""
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 4, 32, 32, generator=g)
    target = torch.randint(0, 4, (2, 32, 32), generator=g)
    out = FocalLossDevice(2.0, 1.0, True)(logits.cuda(), target.cuda())
    assert int(out["bad_targets"]) == 0
    bad = target.clone()
    bad[0, 0, :5] = 7
    bad[1, 3, 3] = -3
    bad[1, 4, 4] = -100                                     # the ignore label is not "bad"
    out2 = FocalLossDevice(2.0, 1.0, True)(logits.cuda(), bad.cuda())
    assert int(out2["bad_targets"]) == 6
    assert torch.isfinite(out2["loss"]).all() and float(out2["dlogits"][0, :, 0, :5].abs().max()) == 0.0
