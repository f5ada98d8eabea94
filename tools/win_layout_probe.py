"""Is the SAM window attention bound by DRAM access granularity?  The same kernel, the same number of (window, head) items and the same
bytes, two layouts of the q | k | v rows: 12 heads per token row (a window-head's K rows are 128-byte pieces 4608 bytes apart) against one
head per token row (E = 64: a window row of 14 tokens is one 5.4 KiB run)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dbglib
_dbglib.use_env_library() if os.environ.get("LA_TOOLS_LIB") else _dbglib.use_debug_library()
import torch
from labelanything_amd import _lib as L

g = torch.Generator(device="cuda").manual_seed(3)
ih, gg, sc = 64, 14, 0.125
nw = -(-ih // gg)


def bench(fn, it=6):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(5):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e_.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e_) / it * 1e3)
    return sorted(ts)[len(ts) // 2]


for nimg, heads in ((96, 12), (1152, 1)):
    e = heads * 64
    b, t, tpad = nimg * nw * nw, gg * gg, (16 * gg + 63) // 64 * 64
    qkv = (torch.randn(nimg * ih * ih, 3 * e, device="cuda", generator=g) * 0.8).half()
    padrow = (torch.randn(3 * e, device="cuda", generator=g) * 0.5).half()
    tabh = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    tabw = (torch.randn(2 * gg - 1, 64, device="cuda", generator=g) * 0.3).half()
    out = torch.empty(nimg * ih * ih, e, dtype=torch.float16, device="cuda")
    us = bench(lambda: L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, sc, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, img_hw=(ih, ih), padrow=padrow))
    gb = (qkv.numel() + out.numel()) * 2 / 1e9
    print(f"{nimg:5d} images x {heads:2d} heads: {us:9.1f} us  {gb / us * 1e3:6.2f} TB/s of q | k | v read once + out written")
