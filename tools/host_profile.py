#!/usr/bin/env python
"""cProfile of the eager (non-graph) host path on the GPU box: where does the per-launch time go?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd.config import LamConfig
from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
cfg = LamConfig(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False)
lam = Lam(cfg, seed=2).cuda()
batch = make_episode(batch=2, n_ways=1, k_shots=1, image_size=1024, seed=1234, prompts=("mask",))
b = {k: (v.cuda() if k in ("images", "prompt_masks") else v) for k, v in batch.items()}
for _ in range(3): lam(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): lam(b)
t_host = (time.perf_counter() - t0) / 5       # host time to ENQUEUE (no sync)
torch.cuda.synchronize()
print(f"host enqueue time per forward: {t_host*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): lam(b)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
