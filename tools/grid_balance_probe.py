#!/usr/bin/env python
"""Persistent four-wave la_gemm with fewer workgroups than CUs (LA_W4_GRID, measurement library): does a BALANCED grid - every workgroup
walks the same number of tiles, e.g. 552 tiles on 184 workgroups x 3 instead of 256 x 2 + 40 - beat the full grid on the HF encoder
shapes (46852 / 57664 rows), whose last round is mostly empty?  One process per grid value (the override is read once):
    for g in 0 184 192 ...; do LA_W4_GRID=$g python tools/grid_balance_probe.py; done"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library
use_debug_library()
import torch
from labelanything_amd import _lib as L

dt = torch.float16
grid = os.environ.get("LA_W4_GRID", "0")
rows = [int(v) for v in os.environ.get("ROWS", "46852,57664").split(",")]
for m in rows:
    for n, k, form in ((768, 3072, "res"), (768, 768, "res"), (768, 2304, "plain32"), (2304, 768, "plain16"), (3072, 768, "gelu")):
        a = torch.randn(m, k, device="cuda").to(dt)
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(dt)
        bias = torch.randn(n, device="cuda")
        o16 = torch.empty(m, n, device="cuda", dtype=dt)
        o32 = torch.zeros(m, n, device="cuda")

        def run():
            if form == "res":
                L.gemm(a, w, bias=bias, res=o32, out32=o32)
            elif form == "plain32":
                L.gemm(a, w, bias=bias, out32=o32)
            elif form == "gelu":
                L.gemm(a, w, bias=bias, out16=o16, act=L.ACT_GELU)
            else:
                L.gemm(a, w, bias=bias, out16=o16)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(30):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 30 * 1e3
        tiles = ((m + 255) // 256) * (n // 256)
        print(f"grid {grid:>4s}  {m:6d} x {n:4d} x {k:4d} {form:8s} tiles {tiles:5d}  {us:8.1f} us  {2 * m * n * k / us / 1e6:7.1f} TF/s", flush=True)
