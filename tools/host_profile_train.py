#!/usr/bin/env python
"""cProfile of one cfg3 training step on the GPU box (frozen / trainable encoder): host time per step and where it goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from labelanything_amd.train import LamTrainer
enc = "--train-encoder" in sys.argv
lam, cfg = bench.build_model(torch.float16, torch.float32, "cfg3_train", None)
lam = lam.cuda()
batch = bench.make_inputs(2, 1234, torch.device("cuda"), "cfg3_train")
tr = LamTrainer(lam, lr=5e-5, num_warmup_steps=1000, train_encoder=enc)
gt = torch.randint(0, batch["flag_examples"].shape[2], (2, 480, 480)).cuda()
for _ in range(2):
    tr.step(batch, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tr.step(batch, gt)
t_host = (time.perf_counter() - t0) / 3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 3
print(f"train_encoder={enc}: host enqueue {t_host * 1e3:.1f} ms per step, wall {t_all * 1e3:.1f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(batch, gt)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
