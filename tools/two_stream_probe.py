#!/usr/bin/env python
"""Probe: does running two half-batches concurrently on two HIP streams (two graphs) beat one full batch?
Complementary kernels (MFMA-bound GEMM, exp-bound attention, HBM-bound LayerNorm) could overlap across the halves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, make_inputs


def run(n_models, episodes_each, steps=20, warm=3):
    dev = torch.device("cuda", 0)
    lams, batches, streams = [], [], []
    for i in range(n_models):
        lam, _ = build_model(torch.float16, torch.float32)
        lam = lam.to(dev)
        lam.use_graphs = True
        lams.append(lam)
        batches.append(make_inputs(episodes_each, 1234 + i, dev))
        streams.append(torch.cuda.Stream())
    def step():
        for lam, b, s in zip(lams, batches, streams):
            with torch.cuda.stream(s):
                lam(b)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return n_models * episodes_each / dt, dt * 1e3


if __name__ == "__main__":
    for n, e in ((1, 8), (2, 4), (2, 8), (4, 2), (1, 16)):
        eps, ms = run(n, e)
        print(f"{n} stream(s) x {e} episodes: {eps:7.1f} episodes/s  ({ms:.2f} ms per round)", flush=True)
