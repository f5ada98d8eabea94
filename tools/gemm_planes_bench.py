#!/usr/bin/env python
"""Timing + correctness of la_gemm on the encoder GEMM shapes of one 16-episode cfg2 step, single-plane and two-plane
(split-precision) weights, per kernel path (LA_GEMM_PATH is read once per process, so every path runs in a subprocess).

    python tools/gemm_planes_bench.py            # all paths
    python tools/gemm_planes_bench.py --one      # the path selected by the environment only
"""
import math
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_debug_library  # noqa: E402

use_debug_library()      # LA_GEMM_PATH is honoured by the -DLA_DEBUG library only

SHAPES = [(131072, 3072, 768, "lin1", 1), (131072, 768, 3072, "lin2", 2), (131072, 2304, 768, "qkv g", 2),
          (156800, 2304, 768, "qkv w", 2), (131072, 768, 768, "proj", 2), (131072, 768, 3072, "lin2", 1),
          (131072, 2304, 768, "qkv g", 1), (131072, 768, 768, "proj", 1), (4096, 4096, 4096, "4k cube", 1), (4096, 4096, 4096, "4k cube", 2)]


def one():
    import torch
    from labelanything_amd import _lib as L
    from tools.bench_ops import timeit
    dt = torch.float16
    tag = os.environ.get("LA_GEMM_PATH", "auto")
    for m, n, k, name, planes in SHAPES:
        a = torch.randn(m, k, device="cuda").to(dt)
        w32 = torch.randn(n, k, device="cuda") / math.sqrt(k)
        hi = w32.to(dt)
        w = torch.cat([hi, (w32 - hi.float()).to(dt)], dim=1).contiguous() if planes == 2 else hi
        bias = torch.randn(n, device="cuda")
        o16 = torch.empty(m, n, device="cuda", dtype=dt)
        o32 = torch.empty(m, n, device="cuda")
        kw = dict(a_kmod=k) if planes == 2 else {}
        try:
            L.gemm(a, w, bias=bias, out32=o32, **kw)
            rows = slice(0, 4096)
            ref = a[rows].float() @ (w32 if planes == 2 else hi.float()).t() + bias
            err = float((o32[rows] - ref).abs().max() / ref.abs().max())
            t = timeit(lambda: L.gemm(a, w, bias=bias, out16=o16, **kw))
        except RuntimeError as ex:
            print(f"[{tag}] {name:8s} planes={planes}: {ex}")
            continue
        print(f"[path {tag:4s}] {name:8s} {m}x{n}x{k} planes={planes}: {t*1e6:8.1f} us  algorithmic {2*m*n*k/t/1e12:7.1f} TF/s  "
              f"issued {2*m*n*k*planes/t/1e12:7.1f} TF/s  err {err:.2e}", flush=True)
        del a, w, o16, o32


if __name__ == "__main__":
    if "--one" in sys.argv:
        sys.argv.remove("--one")
        one()
    else:
        for path in (sys.argv[1:] or ["", "7", "6", "4", "2"]):
            env = dict(os.environ)
            if path and path != "auto":
                env["LA_GEMM_PATH"] = path
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)
