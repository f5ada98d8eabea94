#!/usr/bin/env python
"""Comparator for the north_star's ">= 6x the reference single-GPU PyTorch-eager" target: the oracle
(the checked restatement of the reference's eager torch op sequence) run ON the MI355X with stock torch/rocBLAS/MIOpen
kernels, fp32 and fp16-autocast, next to the HIP path.  A measurement script (not a pytest test); lives under tests/
because it executes the oracle.      python tests/perf_eager_gpu.py [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd.config import LamConfig
from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from labelanything_amd.weights import init_state_dict
from oracle import lam_oracle as O
from tests.cases import geometry_for


def timed(fn, warm=2, it=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


def main():
    episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = LamConfig(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False)
    sd = {k: v.cuda() for k, v in init_state_dict(cfg, 2).items()}
    geo = geometry_for(cfg)
    batch = make_episode(batch=episodes, n_ways=1, k_shots=1, image_size=1024, seed=1234, prompts=("mask",))
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    # the oracle builds a few helper tensors on the default device
    torch.set_default_device("cuda")
    with torch.no_grad():
        t32 = timed(lambda: O.lam_forward(sd, geo, dev_batch))
        with torch.autocast("cuda", dtype=torch.float16):
            try:
                t16 = timed(lambda: O.lam_forward(sd, geo, dev_batch))
            except Exception as e:   # the reference itself breaks under plain half casts; autocast usually works
                t16 = None
                print("fp16 autocast failed:", type(e).__name__, e)
    torch.set_default_device("cpu")
    lam = Lam(cfg, seed=2).cuda()
    lam.use_graphs = True
    th = timed(lambda: lam(dev_batch), warm=3, it=10)
    print(f"episodes per forward: {episodes}")
    print(f"torch eager fp32 on MI355X : {episodes / t32:8.2f} episodes/s")
    if t16:
        print(f"torch eager fp16-autocast  : {episodes / t16:8.2f} episodes/s")
    print(f"HIP path (f16 enc, f32 dec): {episodes / th:8.2f} episodes/s   ({t32 / th:.1f}x fp32 eager" + (f", {t16 / th:.1f}x fp16 eager)" if t16 else ")"))


if __name__ == "__main__":
    main()
