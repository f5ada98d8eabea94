#!/usr/bin/env python
"""Run a few launches of one GEMM / attention shape (for rocprofv3 --pmc passes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "lin2"
dt = torch.float16
shapes = {"lin2": (65536, 768, 3072), "lin1": (65536, 3072, 768), "qkv": (65536, 2304, 768)}      # 16-image encoder batch
if which in shapes:
    m, n, k = shapes[which]
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(dt)
    bias = torch.randn(n, device="cuda")
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    for _ in range(5):
        L.gemm(a, w, bias=bias, out16=o16)
else:
    b, heads, g = 8, 12, 64
    t_ = g * g; e = heads * 64
    qkv = torch.randn(b * t_, 3 * e, device="cuda").to(dt)
    vt = torch.randn(b * heads, 64, t_, device="cuda").to(dt)
    out = torch.empty(b * t_, e, device="cuda", dtype=dt)
    tab = torch.randn(2 * g - 1, 64, device="cuda").to(dt)
    for _ in range(5):      # global SAM attention with the rel-pos terms computed in the kernel (the product path)
        L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, t_, g, e, 0.125, L.ATTN_RELPOS, tabh=tab, tabw=tab)
torch.cuda.synchronize()
