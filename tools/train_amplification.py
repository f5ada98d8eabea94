#!/usr/bin/env python
"""Does the decoder-side gradient error of the trainable-encoder fixtures scale with the encoder FORWARD's error (the claim behind their
loose bounds)?  The first step of each reference training fixture (tests/golden/train_step_encoder.*, train_step_sam.*) under encoder
numerics of decreasing forward error - plain 16-bit operands, the training default, every weight as two planes - and the resulting
errors: logits (forward), per-tensor gradient norms and entry-wise gradients, split into encoder tensors and decoder-side tensors."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import load_file
from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from labelanything_amd.train import LamTrainer
from tests.cases import TRAIN_ENC_CASE, TRAIN_SAM_CASE
from tests.helpers import GOLDEN, rel_err

FULL = ("patch", "qkv", "proj", "lin1", "lin2", "neck")


def first_step(case, gold, keys, precise):
    batch = make_episode(**case["episode"])
    kw = {} if precise is None else {"precise": precise}
    lam = Lam(case["cfg"], seed=case["weight_seed"], **kw).cuda()
    tr = LamTrainer(lam, lr=case["lr"], weight_decay=case["weight_decay"], num_warmup_steps=case["warmup"], train_encoder=True)
    tr.zero_grad()
    res = tr.forward_backward(batch, gold["gt"])
    g0 = {k: gv.clone() for k, gv in zip(tr.names, tr.opt.grad_views)}
    fwd = rel_err(res["logits"], gold["logits0"])
    gn = torch.stack([g0[k].norm() for k in keys]).cpu()
    floor_g = 1e-2 * float(gold["grad_norm"].max())
    rel_n = (gn - gold["grad_norm"]).abs() / gold["grad_norm"].clamp_min(floor_g)
    enc = torch.tensor([k.startswith("image_encoder.") for k in keys])
    gmax = max(float(v.abs().max()) for k, v in gold.items() if k.startswith("grad."))
    ent = {"enc": 0.0, "dec": 0.0}
    for k, v in gold.items():
        if k.startswith("grad."):
            err = float((g0[k[5:]].cpu() - v).abs().max()) / max(float(v.abs().max()), 1e-2 * gmax)
            side = "enc" if k[5:].startswith("image_encoder.") else "dec"
            ent[side] = max(ent[side], err)
    return dict(precise=tr.train_precise, forward=fwd, norm_enc=float(rel_n[enc].max()), norm_dec=float(rel_n[~enc].max()),
                entry_enc=ent["enc"], entry_dec=ent["dec"], loss=float(res["loss"]), loss_ref=float(gold["loss"][0]))


for name, case, stem in (("hf_tiny", TRAIN_ENC_CASE, "train_step_encoder"), ("sam_tiny", TRAIN_SAM_CASE, "train_step_sam")):
    gold = load_file(os.path.join(GOLDEN, stem + ".safetensors"))
    with open(os.path.join(GOLDEN, stem + ".json")) as fh:
        keys = json.load(fh)["keys"]
    for precise in ((), None, FULL):
        r = first_step(case, gold, keys, precise)
        print(f"{name:8s} precise={'+'.join(r['precise']) or 'none':34s} forward {r['forward']:.2e}  loss {r['loss']:.6f} (ref {r['loss_ref']:.6f})  "
              f"grad norms: encoder {r['norm_enc']:.2e} decoder side {r['norm_dec']:.2e}  entry-wise: encoder {r['entry_enc']:.2e} decoder side {r['entry_dec']:.2e}",
              flush=True)
