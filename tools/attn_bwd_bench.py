"""Per-shape timing of the flash-attention backward (la_attn_bwd / la_attn_bwd_relpos) next to its forward.
python tools/attn_bwd_bench.py [shape index]; under `rocprofv3 --kernel-trace --stats` one shape per run splits dQ from dK / dV."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from labelanything_amd import _lib as L

SHAPES = (("hf 52x12 T901", 52, 12, 901, 0), ("sam global 4x12 T4096 G64", 4, 12, 4096, 64), ("sam window 100x12 T196 G14", 100, 12, 196, 14))
sel = [int(sys.argv[1])] if len(sys.argv) > 1 else range(len(SHAPES))
for si in sel:
    name, b, heads, t, g = SHAPES[si]
    e, tpad, bh = heads * 64, (t + 63) // 64 * 64, b * heads
    qkv = (torch.randn(b * t, 3 * e, device="cuda") * 0.5).half()
    out = torch.empty(b * t, e, device="cuda", dtype=torch.half)
    dout = (torch.randn(b * t, e, device="cuda") * 0.1).half()
    dqkv = torch.zeros_like(qkv)
    vt, kt, qt, dot = (torch.zeros(bh, 64, tpad, device="cuda", dtype=torch.half) for _ in range(4))
    lse, dvec = torch.zeros(bh, tpad, device="cuda"), torch.zeros(bh, tpad, device="cuda")
    L.head_transpose(qkv, 2 * e, b, heads, t, tpad, vt)
    L.head_transpose(qkv, e, b, heads, t, tpad, kt)
    L.head_transpose(qkv, 0, b, heads, t, tpad, qt)
    L.head_transpose(dout, 0, b, heads, t, tpad, dot)
    scale = 0.125
    if g:
        relh, relw = torch.randn(bh, t, g, device="cuda") * 0.3, torch.randn(bh, t, g, device="cuda") * 0.3
        drh, drw = torch.empty_like(relh), torch.empty_like(relw)
        fwd = lambda: L.attn_fwd_relpos_lse(qkv, vt, out, relh, relw, lse, b, heads, t, tpad, g, e, scale)
        bwd = lambda: L.attn_bwd_relpos(qkv, out, dout, kt, qt, dot, lse, dvec, dqkv, relh, relw, drh, drw, b, heads, t, tpad, g, e, scale)
    else:
        fwd = lambda: L.attn_fwd_lse(qkv, vt, out, lse, b, heads, t, tpad, e, scale)
        bwd = lambda: L.attn_bwd(qkv, out, dout, kt, qt, dot, lse, dvec, dqkv, b, heads, t, tpad, e, scale)
    res = []
    for f in (fwd, bwd):
        for _ in range(3): f()
        best = 1e9
        for _ in range(3):
            s, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): f()
            e2.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e2) / 10 * 1e3)
        res.append(best)
    fl = 2.0 * t * t * 64 * bh
    print(f"{name}: forward {res[0]:.1f} us ({2 * fl / res[0] / 1e6:.0f} TF/s)  backward {res[1]:.1f} us ({7 * fl / res[1] / 1e6:.0f} TF/s, "
          f"{res[1] / res[0]:.2f} x forward)", flush=True)
