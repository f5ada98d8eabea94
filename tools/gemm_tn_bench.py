"""la_gemm_tn on the weight-gradient shapes of the cfg3 training step (M rows x N x K, fp32)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from labelanything_amd import _lib as L
for m, n, k in ((270000, 128, 256), (270000, 256, 128), (1228800, 256, 16), (1228800, 16, 16), (46800, 256, 2304), (46800, 256, 768),
                (28800, 32, 288), (1800, 128, 256), (300, 256, 256)):
    dy, x, dw = torch.randn(m, n, device="cuda"), torch.randn(m, k, device="cuda"), torch.zeros(n, k, device="cuda")
    f = lambda: L.gemm_tn(dy, x, dw)
    for _ in range(3): f()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    gb = m * (n + k) * 4 / 1e9
    print(f"gemm_tn[{m}x{n}x{k}]: {best:8.1f} us  {2.0 * m * n * k / best / 1e6:6.1f} TF/s  {gb / best * 1e3:5.2f} TB/s", flush=True)
