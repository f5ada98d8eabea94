import math, os, sys
sys.path.insert(0, "/root/repo")
import torch
from labelanything_amd import _lib as L
g = torch.Generator(device="cuda").manual_seed(3)
heads, e, sc = 12, 768, 0.125
nimg, ih, gg = 96, 64, 14
nw = -(-ih // gg)
b, t, tpad = nimg * nw * nw, gg * gg, (16 * gg + 63) // 64 * 64
def bench(fn, it=6):
    for _ in range(2): fn()
    ts = []
    for _ in range(5):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e_.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e_) / it * 1e3)
    return sorted(ts)[len(ts) // 2]
qkv = (torch.randn(nimg * ih * ih, 3 * e, device="cuda", generator=g) * 0.8).half()
padrow = (torch.randn(3 * e, device="cuda", generator=g) * 0.5).half()
tabh = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
tabw = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
out = torch.empty(nimg * ih * ih, e, dtype=torch.float16, device="cuda")
us = bench(lambda: L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, img_hw=(ih, ih), padrow=padrow))
print(f"window14 image order {b}x12: {us:9.1f} us")
qkvw = (torch.randn(b * t, 3 * e, device="cuda", generator=g) * 0.8).half()
outw = torch.empty(b * t, e, dtype=torch.float16, device="cuda")
us = bench(lambda: L.attn_fwd_rows(qkvw, outw, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw))
print(f"window14 window buffers, rows {b}x12: {us:9.1f} us")
vt = torch.zeros(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
us = bench(lambda: L.attn_fwd(qkvw, vt, outw, None, None, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh, tabw))
print(f"window14 window buffers, V^T {b}x12: {us:9.1f} us")
