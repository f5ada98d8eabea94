#!/usr/bin/env python
"""W [N, K] fp32 -> W^T [K, N] fp16 (the 'weight' of a data-gradient GEMM, one copy per encoder weight and optimizer step): the 32 x 32
NCHW -> NHWC transposer against la_transpose16 (64 x 64 tiles, packed 16-bit stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L


def timed(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for n, k in ((768, 3072), (3072, 768), (2304, 768), (768, 768), (1024, 4096), (1280, 5120)):
    w = torch.randn(n, k, device="cuda")
    a = torch.empty(k, n, device="cuda", dtype=torch.float16)
    b = torch.empty(k, n, device="cuda", dtype=torch.float16)
    ta = timed(lambda: L.nchw_to_nhwc(w, 1, n, k, out16=a, dt=L.LA_F16))
    tb = timed(lambda: L.transpose16(w, b))
    print(f"{n:5d} x {k:5d}: nchw_to_nhwc {ta:6.1f} us | transpose16 {tb:6.1f} us | equal {bool(torch.equal(a, b))} | {n * k * 6 / min(ta, tb) / 1e6:.2f} TB/s best")
