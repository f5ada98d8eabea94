"""GPU: the north_star logit tolerance must not depend on the one seed per geometry that the golden fixtures pin.

For the two BASELINE geometries whose encoders run with 16-bit MFMA operands (cfg2 SAM ViT-B 1024, cfg1 ViT-MAE-B 480) other weight
AND episode seeds are drawn, the CPU oracle (pinned on the reference, tools/make_golden.py) runs beside the HIP path, and every
stage is held to 1e-3.  ``python tests/test_parity_seeds_gpu.py [seeds...] [--precise=patch,v,proj,neck]`` prints the same numbers
for other seeds / split-precision group sets (profiles/r0N_parity_seeds.log)."""
import os
import sys

import pytest
import torch

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from labelanything_amd.weights import init_state_dict
from oracle import lam_oracle as O
from tests.cases import CASES, geometry_for
from tests.helpers import argmax_disagreement, rel_err

pytestmark = pytest.mark.gpu

NAMES = ("cfg2_sam_b_1024_1w1s", "cfg1_mae_b_480_1w1s")


def seed_errors(name: str, seed: int, precise="auto", norm_fold=True):
    case = CASES[name]
    cfg = case["cfg"]
    lam = Lam(cfg, seed=seed, precise=precise).cuda()
    lam.norm_fold = norm_fold
    ep = dict(case["episode"])
    ep["seed"] = seed
    batch = make_episode(**ep)
    rows = None
    if cfg.bank_size:                       # RandomMatrixEncoder: fix the rows on both sides
        c = batch["flag_examples"].shape[-1]
        rows = torch.randperm(cfg.bank_size, generator=torch.Generator().manual_seed(seed))[:c]
        lam.selected_rows = rows
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(cfg, seed), geometry_for(cfg), batch, rows)
    out = lam.forward_argmax(batch)
    torch.cuda.synchronize()
    n_diff, n_real = argmax_disagreement(out["logits"], ref["logits"].argmax(1), ref["logits"], margin_rel=2e-3)
    return {"logits": rel_err(out["logits"], ref["logits"]),
            "class_examples_embeddings": rel_err(out["class_examples_embeddings"], ref["class_examples_embeddings"]),
            "argmax_flips": n_diff / out["argmax"].numel(), "argmax_outside_band": n_real}


@pytest.mark.parametrize("seed", [101, 202, 303])
@pytest.mark.parametrize("name", NAMES)
def test_default_numerics_hold_1e3_over_seeds(name, seed):
    e = seed_errors(name, seed)
    print(f"{name} weight/episode seed {seed}: {e}")
    assert e["logits"] <= 1e-3 and e["class_examples_embeddings"] <= 1e-3, e
    # inside the 2e-3 band near-ties may flip: measured 0.04 - 0.56 % of the pixels over these seeds (random-weight models put many
    # pixels close to a tie); outside the band the argmax must be the reference's everywhere
    assert e["argmax_outside_band"] == 0 and e["argmax_flips"] <= 0.0075, e


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    precise = "auto"
    fold = "--no-fold" not in sys.argv[1:]          # the LayerNorm kernels instead of the folded form (LamEngine.norm_fold)
    for a in sys.argv[1:]:
        if a.startswith("--precise="):
            precise = tuple(g for g in a.split("=", 1)[1].split(",") if g)
    print("precise =", precise, "norm_fold =", fold)
    for nm in NAMES:
        for sd in [int(s) for s in args] or [101, 202, 303]:
            print(nm, sd, seed_errors(nm, sd, precise, fold), flush=True)
