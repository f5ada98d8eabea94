#!/usr/bin/env python
"""CPU study (oracle autograd only): how the decoder-side gradients of the trainable-encoder fixtures react to a perturbation of the
encoder output of the size the 16-bit encoder forward has (1e-3 of the largest entry).  The pre-neck embeddings of the fixture's episode
are perturbed by seeded Gaussian noise scaled to a given max-norm; loss and decoder-side gradients come from the oracle's autograd with
those embeddings as input.  Prints, per noise scale and seed, the worst per-tensor gradient-norm error against the unperturbed run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd.episodes import make_episode
from labelanything_amd.weights import init_state_dict
from oracle import lam_oracle as O, loss_oracle as LO
from tests.cases import TRAIN_ENC_CASE, TRAIN_SAM_CASE, geometry_for
from safetensors.torch import load_file
from tests.helpers import GOLDEN

torch.set_num_threads(16)


def grads(case, b2, gt):
    cfg = case["cfg"]
    w = {k: (v.clone().requires_grad_("image_encoder" not in k and "gaussian" not in k) if v.is_floating_point() else v)
         for k, v in init_state_dict(cfg, case["weight_seed"]).items()}
    out = O.lam_forward(w, geometry_for(cfg), b2)
    loss, _ = LO.focal_objective(out["logits"], gt)
    loss.backward()
    return float(loss), {k: v.grad.clone() for k, v in w.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


for name, case, stem in (("hf_tiny", TRAIN_ENC_CASE, "train_step_encoder"), ("sam_tiny", TRAIN_SAM_CASE, "train_step_sam")):
    gold = load_file(os.path.join(GOLDEN, stem + ".safetensors"))
    batch = make_episode(**case["episode"])
    w0 = init_state_dict(case["cfg"], case["weight_seed"])
    with torch.no_grad():
        im = batch["images"]
        b, n = im.shape[:2]
        e = O.encode_images(w0, geometry_for(case["cfg"]), im.flatten(0, 1))
    base = {k: v for k, v in batch.items() if k != "images"}
    base["embeddings"] = e.view(b, n, *e.shape[1:])
    l0, g0 = grads(case, base, gold["gt"])
    gmax = max(float(v.norm()) for v in g0.values())
    for scale in (1e-4, 3e-4, 1e-3, 3e-3):
        res = []
        for seed in range(6):
            noise = torch.randn(e.shape, generator=torch.Generator().manual_seed(100 + seed))
            noise = noise / noise.abs().max() * scale * e.abs().max()
            b2 = dict(base)
            b2["embeddings"] = (e + noise).view(b, n, *e.shape[1:])
            l1, g1 = grads(case, b2, gold["gt"])
            worst = max((abs(float(g1[k].norm()) - float(g0[k].norm())) / max(float(g0[k].norm()), 1e-2 * gmax), k) for k in g0)
            res.append(worst)
        print(f"{name:8s} embedding perturbation {scale:.0e} of max: worst decoder-side gradient-norm change per seed "
              + ", ".join(f"{w:.1e}" for w, _ in res) + f"   (worst tensor: {max(res)[1]})", flush=True)
