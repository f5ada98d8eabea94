#!/usr/bin/env python
"""Run a few launches of one GEMM / attention shape (for rocprofv3 --pmc passes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANT = int(os.environ.get("LA_VARIANT", "-1"), 0)      # la_gemm_variant (values above 2: the measurement library)
if VARIANT > 2:
    from tools._dbglib import use_debug_library
    use_debug_library()
import torch
from labelanything_amd import _lib as L
if VARIANT >= 0:
    L.gemm_variant(VARIANT)
which = sys.argv[1] if len(sys.argv) > 1 else "lin2"
dt = torch.float16
shapes = {"lin2": (131072, 768, 3072), "lin1": (131072, 3072, 768), "qk": (131072, 1536, 768), "v2": (131072, 768, 768), "cube": (8192, 8192, 8192)}   # 32-image encoder batch
FOLD = {"qkv_n": (2304, 768), "lin1_n": (3072, 768), "proj_p": (768, 768), "lin2_p": (768, 3072)}      # round 6: the folded-LayerNorm forms at the model's 96 images
if which in FOLD:
    m = int(os.environ.get("M", 393216))
    n, k = FOLD[which]
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(dt)
    bias = torch.randn(n, device="cuda")
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    if which.endswith("_n"):
        mr = torch.zeros(-(-m // 256) * 256, 2, device="cuda")
        mr[:, 1] = 1.0
        ncol = w.float().sum(1).contiguous()
        for _ in range(5):
            L.gemm(a, w, bias=bias, out16=o16, act=L.ACT_GELU if which == "lin1_n" else L.ACT_NONE, nstat_in=mr, ncol=ncol)
    else:
        res = torch.zeros(m, n, device="cuda")
        part = torch.empty(m, n // 64, 2, device="cuda")
        rvec = torch.randn(m // 4096, n, device="cuda") if which == "proj_p" else None
        for _ in range(5):
            L.gemm(a, w, bias=bias, res=res, out32=res, out16=o16, nstat_out=part, rvec=rvec, rvec_rpg=4096 if rvec is not None else 0)
elif which in ("attn_rows", "attn_win"):
    b, heads, g = 16, 12, 64
    e = heads * 64
    qkv = torch.randn(b * g * g, 3 * e, device="cuda").to(dt)
    out = torch.empty(b * g * g, e, device="cuda", dtype=dt)
    for _ in range(5):      # the product's attention: no V^T copy; global 64 x 64 rel-pos grid / 14 x 14 windows addressed in image order
        if which == "attn_rows":
            tab = torch.randn(2 * g - 1, 64, device="cuda").to(dt)
            L.attn_fwd_rows(qkv, out, b, heads, g * g, g * g, g, e, 0.125, L.ATTN_RELPOS, tabh=tab, tabw=tab)
        else:
            tab = torch.randn(27, 64, device="cuda").to(dt)
            pad = torch.randn(3 * e, device="cuda").to(dt)
            L.attn_fwd_rows(qkv, out, b * 25, heads, 196, 256, 14, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tab, tabw=tab, img_hw=(g, g), padrow=pad)
elif which in shapes:
    m, n, k = shapes[which]
    a = torch.randn(m, k, device="cuda").to(dt)
    w32 = torch.randn(n, k, device="cuda") / math.sqrt(k)
    w = w32.to(dt)
    bias = torch.randn(n, device="cuda")
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    res = torch.zeros(m, n, device="cuda") if which == "lin2" else None
    vt = torch.zeros((m // 4096) * 12, 64, 4096, device="cuda", dtype=dt) if which == "v2" else None
    if which == "v2":
        w = torch.cat([w, (w32 - w.float()).to(dt)], dim=1).contiguous()
    for _ in range(5):                                   # the epilogues the model uses
        if which == "lin1":
            L.gemm(a, w, bias=bias, out16=o16, act=L.ACT_GELU)
        elif which == "lin2":
            L.gemm(a, w, bias=bias, res=res, out32=res)
        elif which == "v2":
            L.gemm(a, w, bias=bias, out16=o16, vt=vt, vt_col0=0, vt_T=4096, vt_Tpad=4096, vt_hd=64, vt_heads=12, a_kmod=k)
        else:
            L.gemm(a, w, bias=bias, out16=o16)
else:
    b, heads, g = 8, 12, 64
    t_ = g * g; e = heads * 64
    qkv = torch.randn(b * t_, 3 * e, device="cuda").to(dt)
    vt = torch.randn(b * heads, 64, t_, device="cuda").to(dt)
    out = torch.empty(b * t_, e, device="cuda", dtype=dt)
    tab = torch.randn(2 * g - 1, 64, device="cuda").to(dt)
    for _ in range(5):      # global SAM attention with the rel-pos terms computed in the kernel (the product path)
        L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, t_, g, e, 0.125, L.ATTN_RELPOS, tabh=tab, tabw=tab)
torch.cuda.synchronize()
