#!/usr/bin/env python
"""Same-process A/B of the la_gemm main loops on the two q | k | v launches of a SAM ViT-B block at the bench's batch (96 images of
64 x 64 tokens): global attention (V^T columns, identity slots) and 14 x 14 windows from image-order tokens (LA_MAP_WINDOW_PART scatter,
V^T in 16-slot window order).  VARIANTS as in tools/gemm_ab.py; results compared bit for bit."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

variants = [int(v, 0) for v in os.environ.get("VARIANTS", "1,2").split(",")]
rounds = int(os.environ.get("ROUNDS", 5))
b, h, ws, e, heads = int(os.environ.get("IMAGES", 96)), 64, 14, 768, 12
dt = torch.float16
nwy = -(-h // ws)
nb, t = b * nwy * nwy, ws * ws
rows, arows = b * h * h, nb * t
tpad = (16 * ws + 63) // 64 * 64
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, e, device="cuda", generator=g).to(dt)
w = (torch.randn(3 * e, e, device="cuda", generator=g) / math.sqrt(e)).to(dt)
bias = torch.randn(3 * e, device="cuda", generator=g) * 0.1
qkv_w = torch.zeros(arows, 3 * e, device="cuda", dtype=dt)
vt_w = torch.zeros(nb * heads, 64, tpad, device="cuda", dtype=dt)
qkv_g = torch.zeros(rows, 3 * e, device="cuda", dtype=dt)
vt_g = torch.zeros(b * heads, 64, h * h, device="cuda", dtype=dt)
cases = {
    "qkv global": (lambda: L.gemm(x, w, bias=bias, out16=qkv_g, vt=vt_g, vt_col0=2 * e, vt_T=h * h, vt_Tpad=h * h, vt_hd=64, vt_heads=heads), (qkv_g, vt_g)),
    "qkv window": (lambda: L.gemm(x, w, bias=bias, out16=qkv_w, vt=vt_w, vt_col0=2 * e, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=ws,
                                  map=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h)), (qkv_w, vt_w)),
}
for name, (fn, outs) in cases.items():
    res, times = {}, {v: [] for v in variants}
    for v in variants:
        L.gemm_variant(v)
        for o in outs:
            o.zero_()
        fn()
        torch.cuda.synchronize()
        res[v] = [o.clone() for o in outs]
    for r in range(rounds):
        for v in variants:
            L.gemm_variant(v)
            s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            s.record()
            for _ in range(3):
                fn()
            e_.record()
            torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e_) / 3 * 1e3)
    line = f"{name:11s} {rows}x{3 * e}x{e}:"
    for v in variants:
        tt = sorted(times[v])
        med = tt[len(tt) // 2]
        line += f"  v{v} {med:7.1f} us (min {tt[0]:7.1f}) {2.0 * rows * 3 * e * e / med / 1e6:7.1f} TF/s"
    eq = all(torch.equal(a_, b_) for v in variants[1:] for a_, b_ in zip(res[variants[0]], res[v]))
    print(line + f"  bitwise-equal {eq}", flush=True)
L.gemm_variant(2)
