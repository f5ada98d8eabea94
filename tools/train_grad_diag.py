#!/usr/bin/env python
"""Per-tensor gradient error of the full-size cfg3 training step (tests/test_train_gpu.py::test_cfg3_train_step_at_full_size) against
the float64 oracle, over repeated runs: which tensors carry the accumulation error, and how much of it moves from run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from labelanything_amd.config import LamConfig
from labelanything_amd.models import Lam
from labelanything_amd.train import LamTrainer
from tests.test_train_gpu import make_episode, make_gt, oracle_grads

wl = bench.WORKLOADS["cfg3_train"]
cfg = LamConfig(**wl["model"])
batch = make_episode(batch=1, seed=31, prompts=("mask", "point"), **wl["episode"])
c = batch["flag_examples"].shape[2]
gt = make_gt(batch, c, seed=5)
rows = torch.tensor([3, 14, 15, 92, 65, 35])
lam = Lam(cfg, seed=5).cuda()
lam.selected_rows = rows
im = batch["images"]
b, n = im.shape[:2]
e = lam.image_encoder(im.flatten(0, 1).cuda()).float().cpu()
tr = LamTrainer(lam)
runs = []
for it in range(int(os.environ.get("RUNS", 3))):
    tr.zero_grad()
    res = tr.forward_backward(batch, gt)
    torch.cuda.synchronize()
    runs.append({k: v.detach().cpu().clone() for k, v in zip(tr.names, tr.opt.grad_views)})
b2 = {k: v for k, v in batch.items() if k != "images"}
b2["embeddings"] = e.view(b, n, *e.shape[1:])
ref_loss, ref_logits, ref_g = oracle_grads({"cfg": cfg, "weight_seed": 5}, b2, gt, rows, dtype=torch.float64)
gmax = max(float(v.abs().max()) for v in ref_g.values())
print("gmax", gmax, "loss", float(res["loss"]), ref_loss)
for it, g in enumerate(runs):
    rowsd = []
    for k, ref in ref_g.items():
        err = float((g[k] - ref).abs().max())
        rowsd.append((err / gmax, err / max(float(ref.abs().max()), 1e-3 * gmax), float(ref.abs().max()) / gmax, k, tuple(ref.shape)))
    rowsd.sort(reverse=True)
    print(f"run {it}: worst abs {rowsd[0][0]:.2e}; top tensors (abs err / gmax, err / own scale, own max / gmax):")
    for r in rowsd[:10]:
        print(f"   {r[0]:.2e} {r[1]:.2e} {r[2]:.2e} {r[3]} {r[4]}")
    if it:
        d = max(float((g[k] - runs[0][k]).abs().max()) / gmax for k in ref_g)
        print(f"   run-to-run difference vs run 0: {d:.2e} of gmax")
