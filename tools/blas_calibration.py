"""Calibration only (never on the product path): what the vendor GEMM (hipBLASLt behind torch.matmul, no epilogue) reaches on
the encoder's shapes, next to la_gemm WITH the epilogue the model uses, so that roofline fractions can be read against a practical
ceiling.  Usage: python tools/blas_calibration.py [--no-persistent]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--no-persistent" in sys.argv:      # an environment override of kernel selection: the -DLA_DEBUG library only
    from tools._dbglib import use_debug_library
    use_debug_library()
    os.environ["LA_GEMM_NO_PERSISTENT"] = "1"
from labelanything_amd import _lib as L  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


# (name, M, N, K, planes, epilogue)
shapes = [("qkv global", 131072, 2304, 768, 2, "vt"), ("qkv window", 156800, 2304, 768, 2, "vtw"), ("proj", 131072, 768, 768, 2, "res"),
          ("lin1", 131072, 3072, 768, 1, "gelu"), ("lin2", 131072, 768, 3072, 1, "res"), ("qkv 1 plane", 131072, 2304, 768, 1, "vt"),
          ("cube", 8192, 8192, 8192, 1, "plain")]
for name, m, n, k, planes, epi in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    w = torch.randn(n, k * planes, device="cuda", dtype=torch.float16) * 0.02
    b = torch.zeros(n, device="cuda")
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    wb = w[:, :k].contiguous()
    t_blas = timeit(lambda: torch.matmul(a, wb.t(), out=out))
    kw = {"a_kmod": k} if planes == 2 else {}
    if epi == "gelu":
        fn = lambda: L.gemm(a, w, bias=b, out16=out, act=L.ACT_GELU, **kw)
    elif epi == "res":
        r = torch.zeros(m, n, device="cuda")
        fn = lambda: L.gemm(a, w, bias=b, res=r, out32=r, **kw)
    elif epi in ("vt", "vtw"):
        t = 4096 if epi == "vt" else 196
        tpad = 4096 if epi == "vt" else 256
        vt = torch.zeros((m // t) * 12, 64, tpad, device="cuda", dtype=torch.float16)
        fn = lambda: L.gemm(a, w, bias=b, out16=out, vt=vt, vt_col0=1536, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=12,
                            vt_ws=0 if epi == "vt" else 14, **kw)
    else:
        fn = lambda: L.gemm(a, w, bias=b, out16=out, **kw)
    t_la = timeit(fn)
    fl = 2.0 * m * n * k
    print(f"{name:12s} {m}x{n}x{k} planes={planes} {epi:5s}: hipBLASLt(1 plane, no epilogue) {t_blas*1e6:8.1f} us {fl/t_blas/1e12:7.1f} TF/s | "
          f"la_gemm {t_la*1e6:8.1f} us  algorithmic {fl/t_la/1e12:7.1f}  issued {fl*planes/t_la/1e12:7.1f} TF/s", flush=True)
