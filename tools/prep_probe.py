import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from labelanything_amd.config import LamConfig
from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
cfg = LamConfig(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False)
lam = Lam(cfg, seed=2).cuda()
for episodes in (2, 4):
    batch = make_episode(batch=episodes, n_ways=1, k_shots=1, image_size=1024, seed=1234, prompts=("mask",))
    dev = {k: v.cuda() for k, v in batch.items()}
    mixed = {k: (v.cuda() if k in ("images", "prompt_masks") else v) for k, v in batch.items()}
    for name, b in (("all-on-device", dev), ("bench-style (flags/dims on host)", mixed)):
        for graphs in (False, True):
            lam.use_graphs = graphs
            for _ in range(3): lam(b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): lam(b)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            # host-only cost of prepare
            t1 = time.perf_counter()
            for _ in range(10): lam._prepare(b)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - t1) / 10
            print(f"ep={episodes} {name:34s} graphs={graphs}: {dt*1e3:7.2f} ms/forward  prepare {tp*1e3:6.2f} ms")
