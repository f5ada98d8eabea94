#!/usr/bin/env python
"""Same-process A/B of the la_gemm main loops (la_gemm_variant 0: BK 32 persistent, 1: BK 64 quadrant phases, 2: four waves x 512
registers; BLAS=1 adds the vendor GEMM without epilogue as a calibration column) on the encoder shapes
with the model's epilogues: interleaved rounds, median / min microseconds, TFLOP/s, and a bitwise comparison of the results."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

M = int(os.environ.get("M", 131072))
SHAPES = [("lin1", M, 3072, 768), ("lin2", M, 768, 3072), ("qk", M, 1536, 768), ("proj1", M, 768, 768), ("sq8192", 8192, 8192, 8192),
          ("lin1_L", 46886, 4096, 1024), ("edge", 23426, 3072, 768)]
if os.environ.get("SHAPES"):       # "name:m:n:k,..." - names starting with lin1 get the GELU epilogue, lin2 / proj1 the residual one
    SHAPES = [(t.split(":")[0], *(int(x) for x in t.split(":")[1:])) for t in os.environ["SHAPES"].split(",")]
rounds = int(os.environ.get("ROUNDS", 7))

variants = [int(v, 0) for v in os.environ.get("VARIANTS", "0,1").split(",")]      # (0x201: 64-deep loop without the atomic residual epilogue - debug library)
if any(v > 2 for v in variants):
    from tools._dbglib import use_debug_library
    use_debug_library()
else:
    from tools._dbglib import use_env_library
    use_env_library()


import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

dt = torch.float16
LDA = None if not os.environ.get("A_HOT") else 0      # A_HOT=1: row stride 0 - every A row is the same (cache-hot) line set


def run(name, a, w, bias, o16, res):
    m = o16.shape[0]
    if name.startswith("lin1"):
        L.gemm(a, w, bias=bias, out16=o16, act=L.ACT_GELU, M=m, lda=LDA)
    elif name.startswith("nr_"):                    # fp32 output without the residual read (what the read burst of lin2 / proj costs)
        L.gemm(a, w, bias=bias, out32=res, M=m, lda=LDA)
    elif name.startswith("lin2") or name.startswith("proj1"):
        L.gemm(a, w, bias=bias, res=res, out32=res, M=m, lda=LDA)
    else:
        L.gemm(a, w, bias=bias, out16=o16, M=m, lda=LDA)


for name, m, n, k in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(m, k, device="cuda", generator=g).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) / math.sqrt(k)).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    res = torch.zeros(m, n, device="cuda") if name.startswith(("lin2", "proj1", "nr_")) else None
    outs, times = {}, {v: [] for v in variants}
    for v in variants:
        L.gemm_variant(v)
        if res is not None:
            res.zero_()
        run(name, a, w, bias, o16, res)
        torch.cuda.synchronize()
        outs[v] = (res if res is not None else o16).clone()
    for r in range(rounds):
        for v in variants:
            L.gemm_variant(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(name, a, w, bias, o16, res)
            s.record()
            for _ in range(3):
                run(name, a, w, bias, o16, res)
            e.record()
            torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e) / 3 * 1e3)
    ref = (a[:512].float() @ w.float().t() + bias)
    if name.startswith("lin1"):
        ref = torch.nn.functional.gelu(ref)
    line = f"{name:8s} {m}x{n}x{k}:"
    for v in variants:
        t = sorted(times[v])
        med = t[len(t) // 2]
        err = float((outs[v][:512].float() - ref).abs().max() / ref.abs().max())
        line += f"  v{v} {med:7.1f} us (min {t[0]:7.1f}) {2.0 * m * n * k / med / 1e6:7.1f} TF/s err {err:.1e}"
    if os.environ.get("BLAS"):
        tb = []
        for r in range(rounds):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.matmul(a, w.t(), out=o16)
            s.record()
            for _ in range(3):
                torch.matmul(a, w.t(), out=o16)
            e.record()
            torch.cuda.synchronize()
            tb.append(s.elapsed_time(e) / 3 * 1e3)
        tb.sort()
        line += f"  blas {tb[len(tb) // 2]:7.1f} us {2.0 * m * n * k / tb[len(tb) // 2] / 1e6:7.1f} TF/s"
    if len(variants) > 1:
        line += f"  bitwise-equal {torch.equal(outs[variants[0]], outs[variants[1]])}"
    print(line, flush=True)
L.gemm_variant(2)
