"""lin1 / qkv / proj / lin2 shapes of the cfg2 step through la_gemm: time per launch (LA_GEMM_GROUP_M of the measurement library selects
the row panels per tile group).  Under `rocprofv3 --pmc FETCH_SIZE` the same script gives the fetch per launch."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from tools._dbglib import use_env_library
use_env_library()
from labelanything_amd import _lib as L
m = 393216
for name, n, k, kw in (("lin1", 3072, 768, dict(act=L.ACT_GELU)), ("qkv", 2304, 768, {}), ("lin2", 768, 3072, {}), ("proj", 768, 768, {})):
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).half()
    bias = torch.randn(n, device="cuda")
    if name in ("lin2", "proj"):
        res = torch.randn(m, n, device="cuda")
        f = lambda: L.gemm(a, w, bias=bias, res=res, out32=res)
    else:
        out = torch.empty(m, n, device="cuda", dtype=torch.half)
        f = lambda: L.gemm(a, w, bias=bias, out16=out, **kw)
    for _ in range(2): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5 * 1e3
    print(f"{name:5s} {m}x{n}x{k}: {t:8.1f} us  {2.0 * m * n * k / t / 1e6:6.0f} TF/s", flush=True)
