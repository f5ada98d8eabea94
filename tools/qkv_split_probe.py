#!/usr/bin/env python
"""What a q|k / v split of the SAM block's qkv launch would buy: the one-launch form (epilogue_wave on every tile: V^T columns in the
same kernel) against [q|k columns as their own launch] + [v columns with the V^T epilogue], global blocks and 14 x 14 windows, at the
bench's batch.  The q|k launch of the window case is timed WITHOUT its row map (the bound of what a row-mapped epilogue_w4 can reach)
and with it (today's epilogue_wave)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

b, h, ws, e, heads = int(os.environ.get("IMAGES", 96)), 64, 14, 768, 12
dt = torch.float16
nwy = -(-h // ws)
nb, t = b * nwy * nwy, ws * ws
rows, arows = b * h * h, nb * t
tpad = (16 * ws + 63) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, e, device="cuda", generator=g).to(dt)
w = (torch.randn(3 * e, e, device="cuda", generator=g) / math.sqrt(e)).to(dt)
wqk, wv = w[: 2 * e].contiguous(), w[2 * e:].contiguous()
bias = torch.randn(3 * e, device="cuda", generator=g) * 0.1
qkv_w = torch.zeros(arows, 3 * e, device="cuda", dtype=dt)
vt_w = torch.zeros(nb * heads, 64, tpad, device="cuda", dtype=dt)
qkv_g = torch.zeros(rows, 3 * e, device="cuda", dtype=dt)
vt_g = torch.zeros(b * heads, 64, h * h, device="cuda", dtype=dt)
wmap = dict(map=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h))
vg = dict(vt=vt_g, vt_T=h * h, vt_Tpad=h * h, vt_hd=64, vt_heads=heads)
vw = dict(vt=vt_w, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=ws)
cases = [
    ("global: one launch (today)", 3 * e, lambda: L.gemm(x, w, bias=bias, out16=qkv_g, vt_col0=2 * e, **vg)),
    ("global: q|k columns, plain", 2 * e, lambda: L.gemm(x, wqk, bias=bias[: 2 * e], out16=qkv_g[:, : 2 * e])),
    ("global: v columns, V^T", e, lambda: L.gemm(x, wv, bias=bias[2 * e:], out16=qkv_g[:, 2 * e:], vt_col0=0, **vg)),
    ("window: one launch (today)", 3 * e, lambda: L.gemm(x, w, bias=bias, out16=qkv_w, vt_col0=2 * e, **vw, **wmap)),
    ("window: q|k columns, row map (epilogue_wave)", 2 * e, lambda: L.gemm(x, wqk, bias=bias[: 2 * e], out16=qkv_w[:, : 2 * e], **wmap)),
    ("window: q|k columns, NO map (bound)", 2 * e, lambda: L.gemm(x, wqk, bias=bias[: 2 * e], out16=qkv_w[:rows, : 2 * e])),
    ("window: v columns, V^T + row map", e, lambda: L.gemm(x, wv, bias=bias[2 * e:], out16=qkv_w[:, 2 * e:], vt_col0=0, **vw, **wmap)),
]
for name, n, fn in cases:
    ts = []
    for r in range(int(os.environ.get("ROUNDS", 5)) + 1):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        s.record()
        for _ in range(3):
            fn()
        e_.record()
        torch.cuda.synchronize()
        if r:
            ts.append(s.elapsed_time(e_) / 3 * 1e3)
    ts.sort()
    print(f"{name:48s} {rows}x{n}x{e}: {ts[len(ts) // 2]:8.1f} us (min {ts[0]:8.1f})  {2.0 * rows * n * e / ts[len(ts) // 2] / 1e6:7.1f} TF/s", flush=True)
