#!/usr/bin/env python
"""Kernel time of the encoder attention forms on the model's shapes (global rel-pos 64 x 64, plain T = 901 / 4096, 14 x 14 windows), with a
checksum of the output: run once per library build (LA_TOOLS_LIB=...) for a same-box A/B (tools/attn_ab.sh)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools._dbglib import use_env_library
use_env_library()
from labelanything_amd import _lib as L


def bench(fn, it=int(os.environ.get("IT", 12))):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / it * 1e3)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(3)
heads, e = 12, 768
sc = 1 / math.sqrt(64)
for name, b, t, mode, gg in (("global_relpos 16x12 T4096", 16, 4096, L.ATTN_RELPOS, 64), ("plain 16x12 T4096", 16, 4096, L.ATTN_PLAIN, 0),
                             ("plain 64x12 T901", 64, 901, L.ATTN_PLAIN, 0), ("window14 400x12 T196", 400, 196, L.ATTN_RELPOS_WIN16, 14)):
    tpad = (t + 63) // 64 * 64 if mode != L.ATTN_RELPOS_WIN16 else (16 * gg + 63) // 64 * 64
    qkv = (torch.randn(b * t, 3 * e, device="cuda", generator=g) * 0.8).half()
    if mode == L.ATTN_RELPOS_WIN16:
        vt = torch.zeros(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
        v = qkv[:, 2 * e:].view(b, gg, gg, heads, 64).permute(0, 3, 4, 1, 2)                   # (b, heads, 64, y, x) -> 16-wide slot rows
        vt.view(b, heads, 64, tpad)[..., :16 * gg].unflatten(-1, (gg, 16))[..., :gg] = v
    else:
        vt = torch.empty(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
        L.head_transpose(qkv, 2 * e, b, heads, t, tpad, vt)
    out = torch.empty(b * t, e, dtype=torch.float16, device="cuda")
    tabh = tabw = None
    if gg:
        tabh = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
        tabw = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    fn = lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t, tpad, gg, e, sc, mode, tabh, tabw)
    try:
        us = bench(fn)
    except Exception as ex:
        print(name, 'failed:', ex)
        continue
    fl = 4.0 * b * heads * t * t * 64
    print(f"{name:28s} {us:9.1f} us {fl / us / 1e6:7.1f} TF/s  checksum {float(out.float().abs().sum()):.6e}", flush=True)
    if hasattr(L.lib(), "la_attn_fwd_rows"):       # the same launch without the V^T copy (V tiles row-major + LDS transpose reads)
        out2 = torch.empty_like(out)
        fn2 = lambda: L.attn_fwd_rows(qkv, out2, b, heads, t, tpad, gg, e, sc, mode, tabh=tabh, tabw=tabw)
        try:
            us2 = bench(fn2)
            print(f"{(name + ' ROWS')[:28]:28s} {us2:9.1f} us {fl / us2 / 1e6:7.1f} TF/s  checksum {float(out2.float().abs().sum()):.6e}  equal {bool(torch.equal(out, out2))}", flush=True)
        except Exception as ex:
            print(name, 'rows failed:', ex)
# SAM windows addressed in image order (96 images would be the bench's batch: 16 here): 64 x 64 tokens, 5 x 5 windows of 14 x 14
if hasattr(L.lib(), "la_attn_fwd_rows"):
    nimg, ih, gg = 16, 64, 14
    nw = -(-ih // gg)
    b, t, tpad = nimg * nw * nw, gg * gg, (16 * gg + 63) // 64 * 64
    qkv = (torch.randn(nimg * ih * ih, 3 * e, device="cuda", generator=g) * 0.8).half()
    padrow = (torch.randn(3 * e, device="cuda", generator=g) * 0.5).half()
    tabh = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    tabw = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    out = torch.empty(nimg * ih * ih, e, dtype=torch.float16, device="cuda")
    us = bench(lambda: L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, img_hw=(ih, ih), padrow=padrow))
    print(f"{'window14 image order 400x12':28s} {us:9.1f} us {4.0 * b * heads * t * t * 64 / us / 1e6:7.1f} TF/s  checksum {float(out.float().abs().sum()):.6e}", flush=True)
