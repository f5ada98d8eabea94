#!/usr/bin/env python
"""Ablation timing of the four-wave persistent GEMM (gemm_t256w, la_gemm_variant 2) - debug library only, results of the ablated
builds are wrong on purpose: bit 1 no LDS-DMA pieces in the loop, 2 no fragment reads, 4 no waits / barriers, 8 no MFMAs."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library
use_debug_library()
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

SHAPES = [("cube", 8192, 8192, 8192), ("qk", 131072, 1536, 768)]
ABLS = [int(x) for x in os.environ.get("ABLS", "0,1,2,3,4,5,7,8,14").split(",")]
rounds = int(os.environ.get("ROUNDS", 5))
dt = torch.float16
for name, m, n, k in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(m, k, device="cuda", generator=g).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) / math.sqrt(k)).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    times = {v: [] for v in ABLS}
    for r in range(rounds + 1):
        for v in ABLS:
            L.gemm_variant(2 | (v << 12))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.gemm(a, w, bias=bias, out16=o16)
            s.record()
            for _ in range(3):
                L.gemm(a, w, bias=bias, out16=o16)
            e.record()
            torch.cuda.synchronize()
            if r:
                times[v].append(s.elapsed_time(e) / 3 * 1e3)
    for v in ABLS:
        t = sorted(times[v])
        med = t[len(t) // 2]
        print(f"{name:6s} {m}x{n}x{k} abl {v:2d}: {med:8.1f} us (min {t[0]:8.1f})  {2.0 * m * n * k / med / 1e6:7.1f} TF/s-equivalent", flush=True)
L.gemm_variant(2)
