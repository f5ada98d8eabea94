#!/usr/bin/env python
"""Phase timeline of the SAM window attention, one workgroup per (window, head, query block) (attn_fwd_kernel<T, 5, 1, true>): s_memtime
stamps of the workgroup in the middle of a 96-image launch (-DLA_DEBUG library)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library  # noqa: E402

use_debug_library()
import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

lib = L.lib()
buf = (C.c_ulonglong * 64)()
g = torch.Generator(device="cuda").manual_seed(3)
heads, e, sc, ih, gg = 12, 768, 0.125, 64, 14
nw = -(-ih // gg)
for nimg in (96, 1):
    b, t, tpad = nimg * nw * nw, gg * gg, (16 * gg + 63) // 64 * 64
    qkv = (torch.randn(nimg * ih * ih, 3 * e, device="cuda", generator=g) * 0.8).half()
    padrow = (torch.randn(3 * e, device="cuda", generator=g) * 0.5).half()
    tabh = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    tabw = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    out = torch.empty(nimg * ih * ih, e, dtype=torch.float16, device="cuda")
    run = lambda: L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, img_hw=(ih, ih), padrow=padrow)
    for _ in range(3):
        run()
    s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e_.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e_) / 10 * 1e3
    lib.la_dbg_win_stamps(C.cast(buf, C.c_void_p))
    nwg = b * heads * 2
    print(f"{nimg} images: {us:.1f} us per launch, {nwg} items = {us * 768 / nwg:.2f} us per item at 768 in flight")
    for w in range(4):
        t = [int(buf[w * 16 + i]) for i in range(13)]
        d = lambda i, j: t[j] - t[i]
        print(f"   wave {w}: index math + q loads + tile 0 issued {d(0, 1)} | tables (q, table loads, 8 MFMAs, bounce) {d(1, 2)} | tile 0 landed + barrier {d(2, 3)} | "
              + " | ".join(f"tile {j}: work {t[8 + j] - (t[3] if j == 0 else t[4 + j - 1])} wait {t[4 + j] - t[8 + j]}" for j in range(4))
              + f" | normalise + store {d(7, 12)} | total {d(0, 12)} cycles")
