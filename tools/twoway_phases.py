#!/usr/bin/env python
"""Phase timeline of twoway_i2t_kernel (workgroup (0, 0), s_memtime stamps of the -DLA_DEBUG library) on the cfg4 shape:
row burst | phase 1 (Q projection) | phase 2 (attention over the tokens) | phase 3 (out_proj) | phase 4 (LayerNorm + store)."""
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
for d, g, hw, nt in ((256, 120, 4096, 2), (256, 120, 4096, 8), (512, 60, 4096, 2)):
    di = d // 2
    gen = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)
    planes = lambda w: (w.half().contiguous(), (w - w.half().float()).half().contiguous())
    x = rnd(g * hw, d)
    wq, wo = rnd(di, d) / 16, rnd(d, di) / 11
    peq, kt, vt = L.twoway_pe_layout(rnd(hw, di)), rnd(g * nt, di), rnd(g * nt, di)
    bo, gamma, beta = rnd(d), 1 + 0.1 * rnd(d), 0.1 * rnd(d)
    run = lambda: L.twoway_i2t(x, planes(wq), peq, kt, vt, planes(wo), bo, gamma, beta, 1e-5, g, hw, nt, 8)
    pq, po = planes(wq), planes(wo)
    run2 = lambda: L.twoway_i2t(x, pq, peq, kt, vt, po, bo, gamma, beta, 1e-5, g, hw, nt, 8)
    for _ in range(3):
        run2()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run2()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    lib.la_dbg_twoway_stamps(C.cast(buf, C.c_void_p))
    nbytes = 2 * x.numel() * 4
    print(f"D={d} groups={g} hw={hw} nt={nt}: {us:.1f} us per launch = {nbytes / us / 1e3:.0f} GB/s of stream ({nbytes / us / 1e3 / 8000:.3f} of 8 TB/s)")
    names = ["row burst + first staging", "phase 1", "phase 2", "phase 3", "phase 4 + stores"]
    for w in range(4):
        t = [int(buf[w * 16 + i]) for i in range(6)]
        print(f"   wave {w}: " + " | ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)) + f" | total {t[5] - t[0]} cycles")
