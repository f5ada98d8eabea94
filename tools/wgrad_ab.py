"""Weight-gradient product dW += dY^T X at the training shapes: la_transpose16 copies + split-K la_gemm (rounds 3 - 4) against la_gemm_tn16
(row-major 16-bit operands, LDS transpose reads).  `python tools/wgrad_ab.py` on the GPU box; prints us per product and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    shapes = [(46852, 768, 768), (46852, 2304, 768), (46852, 768, 3072), (46852, 3072, 768), (16384, 768, 768), (16384, 2304, 768), (16384, 3072, 768),
              (16384, 768, 3072), (93704, 1024, 1024), (93704, 4096, 1024)]
    for r, n, k in shapes:
        g = torch.Generator().manual_seed(r + n)
        dy = torch.randn(r, n, generator=g).half().cuda()
        x = torch.randn(r, k, generator=g).half().cuda()
        rp = (r + 63) // 64 * 64
        dyt = torch.empty(n, rp, dtype=torch.float16, device="cuda")
        xt = torch.empty(k, rp, dtype=torch.float16, device="cuda")
        dw = torch.zeros(n, k, device="cuda")
        db = torch.zeros(n, device="cuda")
        t_tr_dy = timeit(lambda: L.transpose16(dy, dyt, colsum=db))
        t_tr_x = timeit(lambda: L.transpose16(x, xt))
        t_gemm = timeit(lambda: L.gemm(dyt, xt, out32=dw, ksplit=1))
        t_tn = timeit(lambda: L.gemm_tn16(dy, x, dw, db=db))
        fl = 2.0 * r * n * k
        print(f"dW[{n} x {k}] over {r} rows: transposes {t_tr_dy:.1f} + {t_tr_x:.1f} us, split-K la_gemm {t_gemm:.1f} us ({fl / t_gemm * 1e-6:.0f} TF/s) = "
              f"{t_tr_dy + t_tr_x + t_gemm:.1f} us | la_gemm_tn16 {t_tn:.1f} us ({fl / t_tn * 1e-6:.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
