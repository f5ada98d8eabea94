#!/usr/bin/env python
"""Same-process A/B of the folded-LayerNorm GEMM forms (LaGemmEpilogue.nstat_out / nstat_in) against the plain epilogues of the same
shapes, the passes they replace (la_layernorm_g) and the passes they add (la_norm_finalize): interleaved rounds, median microseconds.

    python tools/normfold_ab.py            (M=393216 rows = 96 images of 64 x 64 tokens; LA_TOOLS_LIB=<other build> for a library A/B)
"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools._dbglib import use_env_library
use_env_library()

import torch  # noqa: E402
from labelanything_amd import _lib as L  # noqa: E402

M = int(os.environ.get("M", 393216))
RPG = int(os.environ.get("RPG", 4096))
E, MLP = 768, 3072
rounds = int(os.environ.get("ROUNDS", 7))
g = torch.Generator(device="cuda").manual_seed(1)


def rn(*s, scale=1.0):
    return torch.randn(*s, device="cuda", generator=g) * scale


x16 = rn(M, E).half()
h16 = rn(M, MLP).half()
stream = rn(M, E)
o16 = torch.empty(M, E, device="cuda", dtype=torch.float16)
qkv = torch.empty(M, 3 * E, device="cuda", dtype=torch.float16)
hbuf = torch.empty(M, MLP, device="cuda", dtype=torch.float16)
part = torch.empty(M, E // 64, 2, device="cuda")
xs = torch.zeros(M, 2 * E, device="cuda", dtype=torch.float16)
xs[:, :E] = x16
mr = torch.zeros(-(-M // 256) * 256, 2, device="cuda")
mr[:, 1] = 1.0
rvec = rn(-(-M // RPG), E, scale=0.1)
w_qkv, w_proj = (rn(3 * E, E) / math.sqrt(E)).half(), (rn(E, E) / math.sqrt(E)).half()
w_l1, w_l2 = (rn(MLP, E) / math.sqrt(E)).half(), (rn(E, MLP) / math.sqrt(MLP)).half()
b3, b1, bm = rn(3 * E), rn(E), rn(MLP)
c3, cm = w_qkv.float().sum(1).contiguous(), w_l1.float().sum(1).contiguous()
gam, bet = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
xpart = torch.empty((M // RPG) * L.ln_cs_chunks(RPG) * E, device="cuda") if M % RPG == 0 else None

CASES = {
    "proj  plain (EPI 3)": lambda: L.gemm(x16, w_proj, bias=b1, res=stream, out32=stream),
    "proj  producer": lambda: L.gemm(x16, w_proj, bias=b1, res=stream, out32=stream, out16=o16, nstat_out=part, rvec=rvec, rvec_rpg=RPG),
    "proj  producer, no rvec": lambda: L.gemm(x16, w_proj, bias=b1, res=stream, out32=stream, out16=o16, nstat_out=part),
    "proj  producer, plane-pair stream in place": lambda: L.gemm(x16, w_proj, bias=b1, out16=xs[:, :E], aux16=xs[:, E:], nstat_out=part, rvec=rvec, rvec_rpg=RPG),
    "lin2  plain (EPI 3)": lambda: L.gemm(h16, w_l2, bias=b1, res=stream, out32=stream),
    "lin2  producer": lambda: L.gemm(h16, w_l2, bias=b1, res=stream, out32=stream, out16=o16, nstat_out=part),
    "lin2  producer, plane-pair stream in place": lambda: L.gemm(h16, w_l2, bias=b1, out16=xs[:, :E], aux16=xs[:, E:], nstat_out=part),
    "qkv   plain (EPI 1)": lambda: L.gemm(x16, w_qkv, bias=b3, out16=qkv),
    "qkv   consumer": lambda: L.gemm(x16, w_qkv, bias=b3, out16=qkv, nstat_in=mr, ncol=c3),
    "qkv   consumer, operand = hi plane (row stride 2 E)": lambda: L.gemm(xs[:, :E], w_qkv, bias=b3, out16=qkv, nstat_in=mr, ncol=c3),
    "lin1  plain (EPI 2)": lambda: L.gemm(x16, w_l1, bias=bm, out16=hbuf, act=L.ACT_GELU),
    "lin1  consumer": lambda: L.gemm(x16, w_l1, bias=bm, out16=hbuf, act=L.ACT_GELU, nstat_in=mr, ncol=cm),
    "layernorm_g (replaced)": lambda: L.layernorm_g(stream, rvec, RPG, gam, bet, 1e-6, out16=o16),
    "norm_finalize": lambda: L.norm_finalize(part, M, E, 1e-6, mr),
}
if xpart is not None:
    CASES["layernorm_g + column sums (replaced)"] = lambda: L.layernorm_g(stream, rvec, RPG, gam, bet, 1e-6, out16=o16, colsum_part=xpart)
    CASES["norm_finalize + column sums"] = lambda: L.norm_finalize(part, M, E, 1e-6, mr, x16=x16, rpg=RPG, cs_part=xpart)

times = {k: [] for k in CASES}
for f in CASES.values():
    f()
torch.cuda.synchronize()
for r in range(rounds):
    for k, f in CASES.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f()
        s.record()
        for _ in range(3):
            f()
        e.record()
        torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) / 3 * 1e3)
print(f"M = {M} rows, rows per group {RPG}, library {L.LIB_PATH}")
for k, t in times.items():
    t.sort()
    print(f"{k:40s} {t[len(t) // 2]:8.1f} us (min {t[0]:8.1f})")
