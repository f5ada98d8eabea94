#!/usr/bin/env python
"""la_layernorm_g at the cfg2 shape (96 images x 4096 tokens x 768): plain, with the per-image vector, with the column sums of the output
(rows per partial = LA_LN_CS_ROWS of the library build: LN_CS_ROWS=<n> here must match the library given by LA_TOOLS_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools._dbglib import use_env_library
use_env_library()
from labelanything_amd import _lib as L
L.LN_CS_ROWS = int(os.environ.get("LN_CS_ROWS", L.LN_CS_ROWS))
bn, hw, e = int(os.environ.get("IMAGES", 96)), int(os.environ.get("HW", 4096)), 768
x = torch.randn(bn * hw, e, device="cuda")
rv = torch.randn(bn, e, device="cuda")
g, b = torch.ones(e, device="cuda"), torch.zeros(e, device="cuda")
y16 = torch.empty(bn * hw, e, device="cuda", dtype=torch.float16)
part = torch.empty(bn * L.ln_cs_chunks(hw) * e, device="cuda")


def bench(fn, it=10):
    for _ in range(3):
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


for name, fn in (("plain", lambda: L.layernorm(x, g, b, 1e-6, out16=y16)),
                 ("+ per-image vector", lambda: L.layernorm_g(x, rv, hw, g, b, 1e-6, out16=y16)),
                 ("+ vector + column sums", lambda: L.layernorm_g(x, rv, hw, g, b, 1e-6, out16=y16, colsum_part=part))):
    print(f"LN_CS_ROWS {L.LN_CS_ROWS:4d}  {name:26s} {bench(fn):8.1f} us", flush=True)
ref = torch.empty(bn, e, device="cuda")
L.colsum_fold(part, bn, L.ln_cs_chunks(hw), e, 1.0 / hw, ref)
print("mean check", float((ref - y16.float().view(bn, hw, e).mean(1)).abs().max()))
