"""ctypes binding of libla_hip.so (include/la_hip.h).

The library is the product: there is NO fallback.  If the shared object is missing or a call
fails, a RuntimeError is raised (allocation failures keep the substring "out of memory" that the
reference's training loop greps for, /root/reference/label_anything/experiment/run.py:339-340).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libla_hip.so")

LA_F16, LA_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
MAP_NONE, MAP_GROUP, MAP_WINDOW_MERGE, MAP_CONVT2X2 = 0, 1, 2, 3
ATTN_PLAIN, ATTN_RELPOS = 0, 1

_DT = {torch.float16: LA_F16, torch.bfloat16: LA_BF16}


class LaGemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_int), ("res_mod", C.c_int),
        ("out32", C.c_void_p), ("ld32", C.c_int), ("out16", C.c_void_p), ("ld16", C.c_int),
        ("act", C.c_int), ("map", C.c_int),
        ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int), ("p3", C.c_int), ("p4", C.c_int),
        ("vt", C.c_void_p), ("vt_col0", C.c_int), ("vt_T", C.c_int), ("vt_Tpad", C.c_int),
        ("vt_hd", C.c_int), ("vt_heads", C.c_int),
    ]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"labelanything_amd: HIP extension {LIB_PATH} is missing - build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU/eager fallback)")
        l = C.CDLL(LIB_PATH)
        l.la_last_error.restype = C.c_char_p
        l.la_version.restype = C.c_int
        for name in EXPORTS:
            getattr(l, name).restype = C.c_int
        _lib = l
    return _lib


# every symbol include/la_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "la_gemm", "la_layernorm", "la_im2col_patch", "la_im2col_3x3", "la_relpos_terms", "la_attn_fwd",
]


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().la_last_error().decode()
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt_of(t: torch.Tensor) -> int:
    return _DT[t.dtype]


def _dev(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("labelanything_amd kernels need device tensors (no CPU fallback)")


# ----------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, *, bias=None, res=None, res_mod=0, out32=None, out16=None,
         act=ACT_NONE, map=MAP_NONE, p=(0, 0, 0, 0, 0), vt=None, vt_col0=0, vt_T=0, vt_Tpad=0, vt_hd=64,
         vt_heads=0, M=None, lda=None) -> None:
    """C = epilogue(a @ w.T).  a: [M,K] 16-bit (row stride lda), w: [N,K] 16-bit."""
    _dev(a)
    m = a.shape[0] if M is None else M
    k = w.shape[1]
    n = w.shape[0]
    e = LaGemmEpilogue()
    e.bias = bias.data_ptr() if bias is not None else None
    e.res = res.data_ptr() if res is not None else None
    e.ldr = res.stride(-2) if res is not None and res.dim() >= 2 else 0
    e.res_mod = res_mod
    e.out32 = out32.data_ptr() if out32 is not None else None
    e.ld32 = out32.stride(-2) if out32 is not None else 0
    e.out16 = out16.data_ptr() if out16 is not None else None
    e.ld16 = out16.stride(-2) if out16 is not None else 0
    e.act, e.map = act, map
    e.p0, e.p1, e.p2, e.p3, e.p4 = p
    e.vt = vt.data_ptr() if vt is not None else None
    e.vt_col0, e.vt_T, e.vt_Tpad, e.vt_hd, e.vt_heads = vt_col0, vt_T, vt_Tpad, vt_hd, vt_heads
    rc = lib().la_gemm(_ptr(a), C.c_int(a.stride(0) if lda is None else lda), _ptr(w), C.c_int(w.stride(0)),
                       C.c_int(m), C.c_int(n), C.c_int(k), C.byref(e), C.c_int(dt_of(a)), _stream())
    _check(rc, "la_gemm")


def layernorm(x: torch.Tensor, gamma, beta, eps: float, *, x2=None, gelu=False, out32=None, out16=None,
              out16_pe=None, pe=None, pe_mod=0, window=0, H=0, W=0, dt=LA_F16) -> None:
    _dev(x)
    rows, e = x.shape[0], x.shape[1]
    rc = lib().la_layernorm(_ptr(x), _ptr(x2), C.c_int(x.stride(0)), C.c_int(rows), C.c_int(e), _ptr(gamma), _ptr(beta),
                            C.c_float(eps), C.c_int(int(gelu)), _ptr(out32), _ptr(out16), _ptr(out16_pe), _ptr(pe),
                            C.c_int(pe_mod), C.c_int(window), C.c_int(H), C.c_int(W), C.c_int(dt), _stream())
    _check(rc, "la_layernorm")


def im2col_patch(img: torch.Tensor, patch: int, out16: torch.Tensor) -> None:
    _dev(img)
    bn, _, s, _ = img.shape
    _check(lib().la_im2col_patch(_ptr(img), C.c_int(bn), C.c_int(s), C.c_int(patch), _ptr(out16), C.c_int(dt_of(out16)),
                                 _stream()), "la_im2col_patch")


def im2col_3x3(x16: torch.Tensor, b: int, h: int, w: int, c: int, out16: torch.Tensor) -> None:
    _dev(x16)
    _check(lib().la_im2col_3x3(_ptr(x16), C.c_int(b), C.c_int(h), C.c_int(w), C.c_int(c), _ptr(out16), C.c_int(dt_of(x16)),
                               _stream()), "la_im2col_3x3")


def relpos_terms(qkv: torch.Tensor, b: int, heads: int, g: int, e: int, tabh, tabw, relh, relw) -> None:
    _check(lib().la_relpos_terms(_ptr(qkv), C.c_int(b), C.c_int(heads), C.c_int(g), C.c_int(e), _ptr(tabh), _ptr(tabw),
                                 _ptr(relh), _ptr(relw), C.c_int(dt_of(qkv)), _stream()), "la_relpos_terms")


def attn_fwd(qkv, vt, out16, relh, relw, b: int, heads: int, t: int, tpad: int, g: int, e: int, scale: float, mode: int) -> None:
    _check(lib().la_attn_fwd(_ptr(qkv), _ptr(vt), _ptr(out16), _ptr(relh), _ptr(relw), C.c_int(b), C.c_int(heads), C.c_int(t),
                             C.c_int(tpad), C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(mode), C.c_int(dt_of(qkv)),
                             _stream()), "la_attn_fwd")
