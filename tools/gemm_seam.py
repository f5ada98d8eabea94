#!/usr/bin/env python
"""Seam timeline of the persistent 64-deep la_gemm kernel (gemm_t256q_kernel), workgroup 0: s_memtime stamps written by the -DLA_DEBUG
library (la_gemm_variant bit 10) at  1 head of a tile's last k-tile | 2 main loop done | 3 epilogue issued | 6 accumulators cleared,
next tile starts | 4 / 5 end of the tile's first / second k-tile.  Prints, per wave group, the cycle deltas between consecutive
stamps averaged over the stamped tiles (the stamps themselves cost ~11 % - read the proportions, not the absolute times).
    make -C labelanything_amd/csrc DEBUG=1 && python tools/gemm_seam.py"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library  # noqa: E402

use_debug_library()
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

NST = 64
dt = torch.float16
M = int(os.environ.get("M", 131072))
SHAPES = [("qk (16-bit epilogue)", M, 1536, 768, "o16", 0), ("lin1 (GELU)", M, 3072, 768, "gelu", 0),
          ("proj (fp32 residual through the slab)", M, 768, 768, "res", 0), ("lin2 (fp32 residual through the slab)", M, 768, 3072, "res", 0)]
lib = L.lib()
lib.la_dbg_gemm_stamps.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * (8 * NST))()
NAMES = {1: "last k-tile head", 2: "main loop done", 3: "epilogue issued", 6: "next tile starts", 4: "k-tile 0 done", 5: "k-tile 1 done"}

for name, m, n, k, kind, vbits in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(m, k, device="cuda", generator=g).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) / math.sqrt(k)).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    o16 = torch.empty(m, n, device="cuda", dtype=dt) if kind != "res" else None
    res = torch.zeros(m, n, device="cuda") if kind == "res" else None

    def run():
        if kind == "gelu":
            L.gemm(a, w, bias=bias, out16=o16, act=L.ACT_GELU)
        elif kind == "res":
            L.gemm(a, w, bias=bias, res=res, out32=res)
        else:
            L.gemm(a, w, bias=bias, out16=o16)

    L.gemm_variant(1 | vbits)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 5 * 1e3
    lib.la_dbg_gemm_stamps_clear()
    L.gemm_variant(1 | vbits | 0x400)
    run()
    lib.la_dbg_gemm_stamps(C.cast(buf, C.c_void_p))
    L.gemm_variant(1)
    tiles_per_cu = math.ceil(math.ceil(m / 256) * (n // 256) / 256)
    print(f"== {name}: {m}x{n}x{k}  {us:.1f} us unstamped = {us / tiles_per_cu:.2f} us per tile ({tiles_per_cu} tiles per CU, {k // 64} k-tiles)")
    for wave in (0, 4):
        ev = [(int(buf[wave * NST + i]) & 0xff, int(buf[wave * NST + i]) >> 8) for i in range(NST) if buf[wave * NST + i]]
        # deltas between consecutive stamps, keyed by (from tag, to tag)
        acc = {}
        for (t0, c0), (t1, c1) in zip(ev, ev[1:]):
            acc.setdefault((t0, t1), []).append(c1 - c0)
        parts = []
        for (t0, t1), v in acc.items():
            v = v[1:] if len(v) > 2 else v            # (drop the first tile: cold caches)
            parts.append(f"{NAMES[t0]} -> {NAMES[t1]}: {sum(v) / len(v):7.0f} cyc (x{len(v)})")
        print(f"   wave {wave} (group {wave // 4}): " + " | ".join(parts))
