#!/usr/bin/env python
"""Seam timeline of the four-wave persistent GEMM (gemm_t256w): s_memtime stamps of workgroup 0 from the measurement library
(la_gemm_variant bit 10): 1 main loop starts | 2 main loop done | 3 epilogue issued.  Environment: LA_W4_GRID (workgroups launched),
LA_W4_STAGGER "P,D" (start classes / delay in 1024-cycle units), SHAPES "name:kind:m:n:k,...", kind in plain / gelu / res."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library
use_debug_library()
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

lib = L.lib()
NST = 128
buf = (C.c_ulonglong * (4 * NST))()
lib.la_dbg_w4_stamps.argtypes = [C.c_void_p]
NAMES = {1: "loop", 2: "done", 3: "epi"}
M = int(os.environ.get("M", 131072))
VB = int(os.environ.get("VBITS", "0"), 0)      # extra la_gemm_variant bits (0x100: no epilogue stores)
SHAPES = [("qk", "plain", M, 1536, 768), ("lin1", "gelu", M, 3072, 768), ("proj", "res", M, 768, 768), ("lin2", "res", M, 768, 3072)]
if os.environ.get("SHAPES"):
    SHAPES = [(t.split(":")[0], t.split(":")[1], *(int(x) for x in t.split(":")[2:])) for t in os.environ["SHAPES"].split(",")]
dt = torch.float16
for name, kind, m, n, k in SHAPES:
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

    L.gemm_variant(2 | VB)
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
    lib.la_dbg_w4_stamps_clear()
    L.gemm_variant(2 | 0x400 | VB)
    run()
    lib.la_dbg_w4_stamps(C.cast(buf, C.c_void_p))
    L.gemm_variant(2)
    grid = min(256, int(os.environ.get("LA_W4_GRID", 256)))
    tiles_per_cu = math.ceil(math.ceil(m / 256) * (n // 256) / grid)
    print(f"== {name}: {m}x{n}x{k} {kind}  {us:.1f} us unstamped = {us / tiles_per_cu:.2f} us per tile ({tiles_per_cu} tiles per workgroup, {k // 64} k-tiles,"
          f" grid {grid}, stagger {os.environ.get('LA_W4_STAGGER', '-')})")
    for wave in (0, 3):
        ev = [(int(buf[wave * NST + i]) & 0xff, int(buf[wave * NST + i]) >> 8) for i in range(NST) if buf[wave * NST + i]]
        acc = {}
        for (t0, c0), (t1, c1) in zip(ev, ev[1:]):
            acc.setdefault((t0, t1), []).append(c1 - c0)
        parts = []
        for (t0, t1), v in acc.items():
            v = v[1:] if len(v) > 2 else v
            parts.append(f"{NAMES[t0]} -> {NAMES[t1]}: {sum(v) / len(v):7.0f} cyc (min {min(v)}, max {max(v)}, x{len(v)})")
        print(f"   wave {wave}: " + " | ".join(parts))
