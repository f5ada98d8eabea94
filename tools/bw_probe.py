#!/usr/bin/env python
"""HBM reference rates at the LayerNorm's shape (393216 x 768): torch's fp32 copy and fp32 -> fp16 cast next to la_layernorm (fp32 in, 16-bit
out) - what a read-4-bytes / write-2-bytes stream reaches on this part."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L

rows, e = 393216, 768
x = torch.randn(rows, e, device="cuda")
y32 = torch.empty_like(x)
y16 = torch.empty(rows, e, device="cuda", dtype=torch.float16)
g, b = torch.ones(e, device="cuda"), torch.zeros(e, device="cuda")


def bench(fn, it=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e_.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e_) / it * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, fn, nbytes in (("torch copy fp32 -> fp32", lambda: y32.copy_(x), 8), ("torch cast fp32 -> fp16", lambda: y16.copy_(x), 6),
                         ("la_layernorm fp32 -> fp16", lambda: L.layernorm(x, g, b, 1e-6, out16=y16), 6),
                         ("torch read-only sum (fp32)", lambda: x.sum(), 4)):
    us = bench(fn)
    print(f"{name:28s} {us:8.1f} us  {rows * e * nbytes / us / 1e6:6.2f} TB/s", flush=True)
