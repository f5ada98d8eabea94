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
LIB_PATH = os.path.join(_HERE, "libla_hip.so")      # the one product library; no environment switch (tools/_dbglib.py re-points this
#                                                     attribute for measurement builds before the first call)

LA_F16, LA_BF16, LA_F32, LA_F16X2 = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD = 0, 1, 2, 3
MAP_NONE, MAP_GROUP, MAP_WINDOW_MERGE, MAP_CONVT2X2, MAP_WINDOW_PART, MAP_CONV3X3 = 0, 1, 2, 3, 4, 5
ATTN_PLAIN, ATTN_RELPOS, ATTN_RELPOS_WIN16 = 0, 1, 2

_DT = {torch.float16: LA_F16, torch.bfloat16: LA_BF16, torch.float32: LA_F32}


class LaGemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_int), ("res_mod", C.c_int),
        ("out32", C.c_void_p), ("ld32", C.c_int), ("out16", C.c_void_p), ("ld16", C.c_int),
        ("act", C.c_int), ("map", C.c_int),
        ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int), ("p3", C.c_int), ("p4", C.c_int),
        ("vt", C.c_void_p), ("vt_col0", C.c_int), ("vt_T", C.c_int), ("vt_Tpad", C.c_int),
        ("vt_hd", C.c_int), ("vt_heads", C.c_int), ("vt_ws", C.c_int), ("amap", C.c_int), ("a_kmod", C.c_int), ("ksplit", C.c_int),
        ("aux16", C.c_void_p), ("ldaux", C.c_int),
        ("nstat_out", C.c_void_p), ("rvec", C.c_void_p), ("rvec_rpg", C.c_int), ("nstat_in", C.c_void_p), ("ncol", C.c_void_p),
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
    "la_dense_pe", "la_point_embed", "la_mask_embed", "la_attn_small", "la_colmean", "la_class_mean",
    "la_classify", "la_add_cast", "la_conv3x3_f32", "la_nchw_to_nhwc", "la_nhwc_to_nchw", "la_bilinear", "la_post_final",
    "la_confmat_update", "la_resample_u8", "la_u8_to_chw_norm", "la_prompt_masks", "la_focal_loss", "la_adamw_step",
    "la_gemm_tn", "la_gemm_tn16", "la_gemm_fused_act_ok", "la_colsum_acc", "la_layernorm_bwd", "la_layernorm_bwd_res", "la_transpose_many", "la_act_fwd", "la_act_bwd", "la_attn_small_lse", "la_attn_small_bwd", "la_bilinear_bwd", "la_bilinear_bwd_set", "la_bilinear_bwd_set_ok", "la_bilinear_rows", "la_bilinear_rows_bwd_set",
    "la_classify_bwd", "la_row_broadcast", "la_twoway_t2i", "la_twoway_i2t", "la_gemm_variant", "la_attn_fwd_lse", "la_head_transpose", "la_attn_bwd", "la_cast", "la_gelu_bwd16", "la_axpy", "la_transpose16", "la_qk_fp8", "la_attn_fwd_fp8", "la_colmean16", "la_layernorm_g", "la_add_rowvec", "la_add_rowvec_split", "la_attn_fwd_cs", "la_attn_fwd_rows", "la_colsum_fold", "la_gelu_fwd16", "la_gemm_tn_db",
    "la_attn_fwd_relpos_lse", "la_attn_bwd_relpos", "la_relpos_bwd", "la_twoway_pe_layout",
    "la_norm_finalize", "la_norm_stats", "la_conv3x3_split", "la_conv3x3_split_ok",
]


def gemm_variant(v: int = -1) -> int:
    """la_gemm_variant: select the main loop of the large encoder GEMMs - 2 (default): four waves x 512 registers (gemm_w4.hip),
    1: eight waves in quadrant phases, 0: the BK 32 kernel.  All are bit-identical.  Returns the previous value."""
    return int(lib().la_gemm_variant(int(v)))


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
    """Every launch goes to the CURRENT device's current stream: refuse tensors that live elsewhere instead of enqueueing a
    kernel on another GPU's stream (single-process multi-GPU hosts must wrap calls in ``torch.cuda.device(t.device)``)."""
    if not t.is_cuda:
        raise RuntimeError("labelanything_amd kernels need device tensors (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "wrap the call in `with torch.cuda.device(tensor.device)`")


# ----------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, *, bias=None, res=None, res_mod=0, out32=None, out16=None,
         act=ACT_NONE, map=MAP_NONE, p=(0, 0, 0, 0, 0), vt=None, vt_col0=0, vt_T=0, vt_Tpad=0, vt_hd=64,
         vt_heads=0, vt_ws=0, M=None, lda=None, amap=MAP_NONE, a_kmod=0, ksplit=0, aux16=None, nstat_out=None, rvec=None, rvec_rpg=0,
         nstat_in=None, ncol=None) -> None:
    """C = epilogue(a @ w.T).  a: [M,K] 16-bit (row stride lda), w: [N,K] 16-bit.  a_kmod > 0: w is [N, j*a_kmod] (split-precision
    planes [W_hi | W_lo]) and the columns of a repeat with period a_kmod.  aux16 (training, shapes with ``gemm_fused_act_ok``): with
    ACT_GELU the pre-activation is written there beside out16 = GELU; with ACT_GELU_BWD out16 = (a @ w.T) * gelu'(aux16).
    nstat_out / rvec / nstat_in / ncol: the producer and consumer sides of a LayerNorm folded into its neighbour GEMMs (see
    LaGemmEpilogue in include/la_hip.h; ``norm_finalize`` turns the producer's partial sums into the consumer's (mean, rstd) rows)."""
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
    e.vt_ws = vt_ws
    e.amap = amap
    e.a_kmod = a_kmod
    e.ksplit = ksplit
    e.aux16 = aux16.data_ptr() if aux16 is not None else None
    e.ldaux = aux16.stride(-2) if aux16 is not None else 0
    if nstat_out is not None or nstat_in is not None or rvec is not None:
        mpad = -(-m // 256) * 256
        if nstat_out is not None and (nstat_out.dtype != torch.float32 or nstat_out.numel() < m * (n // 64) * 2):
            raise RuntimeError(f"gemm: nstat_out must be fp32 with at least M * (N / 64) * 2 = {m * (n // 64) * 2} entries")
        if nstat_in is not None and (nstat_in.dtype != torch.float32 or nstat_in.numel() < mpad * 2 or ncol is None
                                     or ncol.dtype != torch.float32 or ncol.numel() < n):
            raise RuntimeError(f"gemm: nstat_in must be fp32 [ceil(M / 256) * 256, 2] (>= {mpad * 2} entries) and needs ncol fp32 [N]")
        if rvec is not None and (rvec.dtype != torch.float32 or rvec_rpg <= 0 or rvec.numel() < -(-m // rvec_rpg) * n):
            raise RuntimeError("gemm: rvec must be fp32 [ceil(M / rvec_rpg), N] with rvec_rpg > 0")
    e.nstat_out = nstat_out.data_ptr() if nstat_out is not None else None
    e.rvec = rvec.data_ptr() if rvec is not None else None
    e.rvec_rpg = rvec_rpg
    e.nstat_in = nstat_in.data_ptr() if nstat_in is not None else None
    e.ncol = ncol.data_ptr() if ncol is not None else None
    rc = lib().la_gemm(_ptr(a), C.c_int(a.stride(0) if lda is None else lda), _ptr(w), C.c_int(w.stride(0)),
                       C.c_int(m), C.c_int(n), C.c_int(k), C.byref(e), C.c_int(dt_of(a)), _stream())
    _check(rc, "la_gemm")


def norm_finalize(part: Optional[torch.Tensor], m: int, e: int, eps: float, mr: torch.Tensor, x16: Optional[torch.Tensor] = None, rpg: int = 0,
                  cs_part: Optional[torch.Tensor] = None) -> None:
    """mr[row] = (mean, rstd) from a producer GEMM's partial row sums ``part`` ([m, e / 64, 2]; None: mr is an input); cs_part (with x16,
    rpg): column sums of rstd (x16 - mean) per 128-row chunk of every group of rpg rows (``norm_cs_chunks(rpg)`` chunks per group)."""
    _dev(mr)
    mpad = -(-m // 256) * 256
    if mr.dtype != torch.float32 or mr.numel() < mpad * 2 or not mr.is_contiguous():
        raise RuntimeError(f"norm_finalize: mr must be contiguous fp32 with >= {mpad * 2} entries")
    if part is not None and (part.dtype != torch.float32 or part.numel() < m * (e // 64) * 2):
        raise RuntimeError("norm_finalize: part must be fp32 [m, e / 64, 2]")
    if cs_part is not None:
        if x16 is None or rpg <= 0 or m % rpg or x16.shape[0] < m or x16.shape[1] != e or x16.stride(1) != 1:
            raise RuntimeError("norm_finalize: column sums need x16 [m, e] and m % rpg == 0")
        if cs_part.dtype != torch.float32 or cs_part.numel() < (m // rpg) * norm_cs_chunks(rpg) * e:
            raise RuntimeError("norm_finalize: cs_part too small")
    _check(lib().la_norm_finalize(_ptr(part), C.c_int(m), C.c_int(e // 64), C.c_int(e), C.c_float(eps), _ptr(mr), _ptr(x16),
                                  C.c_int(x16.stride(0) if x16 is not None else 0), C.c_int(rpg), _ptr(cs_part),
                                  C.c_int(dt_of(x16) if x16 is not None else LA_F16), _stream()), "la_norm_finalize")


def norm_cs_chunks(rpg: int) -> int:
    """Column-sum partials per group that ``norm_finalize(cs_part=...)`` writes (128 rows each)."""
    return -(-rpg // 128)


def norm_stats(x: torch.Tensor, eps: float, x16: torch.Tensor, mr: torch.Tensor) -> None:
    """x16 = 16-bit rounding of the fp32 rows x, mr[row] = (mean, rstd): the entry of a folded-LayerNorm block stack."""
    _dev(x)
    m, e = x.shape
    mpad = -(-m // 256) * 256
    if x.dtype != torch.float32 or x.stride(1) != 1 or x16.shape != x.shape or not x16.is_contiguous() or mr.dtype != torch.float32 \
            or mr.numel() < mpad * 2:
        raise RuntimeError("norm_stats: x fp32 [m, e], x16 contiguous 16-bit [m, e], mr fp32 with >= ceil(m / 256) * 256 * 2 entries")
    _check(lib().la_norm_stats(_ptr(x), C.c_int(x.stride(0)), C.c_int(m), C.c_int(e), C.c_float(eps), _ptr(x16), _ptr(mr),
                               C.c_int(dt_of(x16)), _stream()), "la_norm_stats")


def gemm_fused_act_ok(m: int, n: int, k: int) -> bool:
    """True when ``gemm(..., aux16=...)`` runs for this shape (the persistent four-wave kernel's direct epilogue)."""
    return bool(lib().la_gemm_fused_act_ok(C.c_int(m), C.c_int(n), C.c_int(k)))


def layernorm(x: torch.Tensor, gamma, beta, eps: float, *, x2=None, gelu=False, out32=None, out16=None,
              out16_pe=None, pe=None, pe_mod=0, window=0, H=0, W=0, dt=LA_F16) -> None:
    _dev(x)
    rows, e = x.shape[0], x.shape[1]
    if window < 0:
        # padded-map forms (LA_MAP_CONV3X3 operand layout): -1 writes the 16-bit output into the interior of [B, H + 2, W + 2] maps, -2 reads
        # the input rows from that layout; ``rows`` counts the UN-padded pixels either way
        o = out32 if out32 is not None else out16
        if window == -2:
            rows = o.shape[0]
            if x.shape[0] < (rows // (H * W)) * (H + 2) * (W + 2):
                raise RuntimeError("layernorm(window=-2): x must hold the padded maps [B * (H + 2) * (W + 2), E]")
        elif out16 is None or out16.shape[0] < (rows // (H * W)) * (H + 2) * (W + 2):
            raise RuntimeError("layernorm(window=-1): out16 must hold the padded maps [B * (H + 2) * (W + 2), ...]")
    rc = lib().la_layernorm(_ptr(x), _ptr(x2), C.c_int(x.stride(0)), C.c_int(rows), C.c_int(e), _ptr(gamma), _ptr(beta),
                            C.c_float(eps), C.c_int(int(gelu)), _ptr(out32), _ptr(out16), _ptr(out16_pe), _ptr(pe),
                            C.c_int(pe_mod), C.c_int(window), C.c_int(H), C.c_int(W), C.c_int(dt), _stream())
    _check(rc, "la_layernorm")


def im2col_patch(img: torch.Tensor, patch: int, out16: torch.Tensor, split: bool = False) -> None:
    """split: out16 is fp16 [rows, 2 K] = [hi | lo] plane pairs (LA_F16X2)."""
    _dev(img)
    bn, _, s, _ = img.shape
    dt = LA_F16X2 if split else dt_of(out16)
    if split and out16.dtype != torch.float16:
        raise RuntimeError("im2col_patch(split=True) writes fp16 plane pairs")
    _check(lib().la_im2col_patch(_ptr(img), C.c_int(bn), C.c_int(s), C.c_int(patch), _ptr(out16), C.c_int(dt),
                                 _stream()), "la_im2col_patch")


def im2col_3x3(x16: torch.Tensor, b: int, h: int, w: int, c: int, out16: torch.Tensor, split: bool = False) -> None:
    """split: x16 rows are fp16 plane pairs [hi (c) | lo (c)], out16 rows [9 taps of hi | 9 taps of lo] (LA_F16X2)."""
    _dev(x16)
    _check(lib().la_im2col_3x3(_ptr(x16), C.c_int(b), C.c_int(h), C.c_int(w), C.c_int(c), _ptr(out16),
                               C.c_int(LA_F16X2 if split else dt_of(x16)), _stream()), "la_im2col_3x3")


def relpos_terms(qkv: torch.Tensor, b: int, heads: int, g: int, e: int, tabh, tabw, relh, relw) -> None:
    _check(lib().la_relpos_terms(_ptr(qkv), C.c_int(b), C.c_int(heads), C.c_int(g), C.c_int(e), _ptr(tabh), _ptr(tabw),
                                 _ptr(relh), _ptr(relw), C.c_int(dt_of(qkv)), _stream()), "la_relpos_terms")


def attn_fwd(qkv, vt, out16, relh, relw, b: int, heads: int, t: int, tpad: int, g: int, e: int, scale: float, mode: int,
             tabh=None, tabw=None) -> None:
    _check(lib().la_attn_fwd(_ptr(qkv), _ptr(vt), _ptr(out16), _ptr(relh), _ptr(relw), _ptr(tabh), _ptr(tabw), C.c_int(b),
                             C.c_int(heads), C.c_int(t),
                             C.c_int(tpad), C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(mode), C.c_int(dt_of(qkv)),
                             _stream()), "la_attn_fwd")


# ---- decoder side ---------------------------------------------------------------------------------
def dense_pe(gauss: torch.Tensor, g: int, d: int, out: torch.Tensor) -> None:
    _dev(gauss)
    _check(lib().la_dense_pe(_ptr(gauss), C.c_int(g), C.c_int(d), _ptr(out), _stream()), "la_dense_pe")


def point_embed(xy, kind, shift, d: int, image_size: int, gauss, type_emb, not_a_point, no_sparse, out32) -> None:
    _dev(xy)
    _check(lib().la_point_embed(_ptr(xy), _ptr(kind), _ptr(shift), C.c_int(kind.numel()), C.c_int(d), C.c_int(image_size),
                                _ptr(gauss), _ptr(type_emb), _ptr(not_a_point), _ptr(no_sparse), _ptr(out32), _stream()),
           "la_point_embed")


def mask_embed(masks, flags, p: int, c: int, hm: int, g: int, d: int, wlist, support, class_enc, pe, src32, src16, srcpe16,
               dt: int) -> None:
    arr = (C.c_void_p * 12)(*[w.data_ptr() for w in wlist])
    _check(lib().la_mask_embed(_ptr(masks), _ptr(flags), C.c_int(p), C.c_int(c), C.c_int(hm), C.c_int(g), C.c_int(d), arr,
                               _ptr(support), _ptr(class_enc), _ptr(pe), _ptr(src32), _ptr(src16), _ptr(srcpe16), C.c_int(dt),
                               _stream()), "la_mask_embed")


def attn_small(q, k, v, b: int, nq: int, nk: int, heads: int, hd: int, *, out16=None, out32=None, dt: int = LA_F16) -> None:
    """q/k/v: fp32 2-D views [rows, ld] (row stride = ld, head h at column h*hd).  dt = LA_F16X2: out16 is [rows, 2 * heads * hd]
    ([hi | lo] planes)."""
    _dev(q)
    o = out16 if out16 is not None else out32
    if dt == LA_F16X2:
        _check(lib().la_attn_small(_ptr(q), C.c_int(q.stride(0)), _ptr(k), C.c_int(k.stride(0)), _ptr(v), C.c_int(v.stride(0)),
                                   C.c_int(b), C.c_int(nq), C.c_int(nk), C.c_int(heads), C.c_int(hd), _ptr(out16), None,
                                   C.c_int(heads * hd), C.c_int(dt), _stream()), "la_attn_small")
        return
    _check(lib().la_attn_small(_ptr(q), C.c_int(q.stride(0)), _ptr(k), C.c_int(k.stride(0)), _ptr(v), C.c_int(v.stride(0)),
                               C.c_int(b), C.c_int(nq), C.c_int(nk), C.c_int(heads), C.c_int(hd), _ptr(out16), _ptr(out32),
                               C.c_int(o.stride(0)), C.c_int(dt), _stream()), "la_attn_small")


COLMEAN_SPLIT = 16


def colmean(x, p: int, hw: int, d: int, out, scratch) -> None:
    """scratch: fp32 [p, COLMEAN_SPLIT, d]."""
    _check(lib().la_colmean(_ptr(x), C.c_int(p), C.c_int(hw), C.c_int(d), _ptr(out), _ptr(scratch), _stream()), "la_colmean")


def class_mean(emb, flags_u8, b: int, m: int, c: int, d: int, out) -> None:
    _check(lib().la_class_mean(_ptr(emb), _ptr(flags_u8), C.c_int(b), C.c_int(m), C.c_int(c), C.c_int(d), _ptr(out), _stream()),
           "la_class_mean")


def classify(feat, protos, b: int, npix: int, c: int, cf: int, seg) -> None:
    _check(lib().la_classify(_ptr(feat), _ptr(protos), C.c_int(b), C.c_int(npix), C.c_int(c), C.c_int(cf), _ptr(seg), _stream()),
           "la_classify")


def add_cast(x, y=None, ymod: int = 0, *, out32=None, out16=None, dt: int = LA_F16) -> None:
    _dev(x)
    rows, d = x.shape[0], x.shape[1]
    _check(lib().la_add_cast(_ptr(x), _ptr(y), C.c_int(ymod), C.c_long(rows), C.c_int(d), _ptr(out32), _ptr(out16), C.c_int(dt),
                             _stream()), "la_add_cast")


def nchw_to_nhwc(x, n: int, c: int, hw: int, *, out32=None, out16=None, dt: int = LA_F16) -> None:
    _dev(x)
    _check(lib().la_nchw_to_nhwc(_ptr(x), C.c_int(n), C.c_int(c), C.c_int(hw), _ptr(out32), _ptr(out16), C.c_int(dt), _stream()),
           "la_nchw_to_nhwc")


def nhwc_to_nchw(x, n: int, c: int, hw: int, out) -> None:
    _dev(x)
    _check(lib().la_nhwc_to_nchw(_ptr(x), C.c_int(n), C.c_int(c), C.c_int(hw), _ptr(out), _stream()), "la_nhwc_to_nchw")


def bilinear(x, n: int, h: int, w: int, oh: int, ow: int, out) -> None:
    _dev(x)
    _check(lib().la_bilinear(_ptr(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(oh), C.c_int(ow), _ptr(out), _stream()),
           "la_bilinear")


def post_final(big, b: int, c: int, s: int, sizes_i32, flag_gts_u8, hmax: int, wmax: int, logits, argmax) -> None:
    _check(lib().la_post_final(_ptr(big), C.c_int(b), C.c_int(c), C.c_int(s), _ptr(sizes_i32), _ptr(flag_gts_u8), C.c_int(hmax),
                               C.c_int(wmax), _ptr(logits), _ptr(argmax), _stream()), "la_post_final")


def confmat_update(pred_i64, gt_i64, lut_i32, k: int, ignore_index: int, confmat_u64, confbin_u64, counters_u64) -> None:
    """pred / gt int64 [B, ...]; lut int32 [B, L] or None; the three accumulators are int64 tensors (bit patterns of u64)."""
    _dev(pred_i64)
    if pred_i64.dtype != torch.int64 or gt_i64.dtype != torch.int64 or pred_i64.shape != gt_i64.shape:
        raise ValueError("confmat_update: pred and gt must be int64 tensors of the same shape")
    if not (pred_i64.is_contiguous() and gt_i64.is_contiguous()):
        raise ValueError("confmat_update: label maps must be contiguous")
    b = pred_i64.shape[0]
    hw = pred_i64.numel() // b
    ell = 0
    if lut_i32 is not None:
        if lut_i32.dtype != torch.int32 or lut_i32.dim() != 2 or lut_i32.shape[0] != b or not lut_i32.is_contiguous():
            raise ValueError("confmat_update: lut must be a contiguous int32 [B, L] tensor")
        ell = lut_i32.shape[1]
    _check(lib().la_confmat_update(_ptr(pred_i64), _ptr(gt_i64), C.c_int(b), C.c_long(hw), _ptr(lut_i32), C.c_int(ell), C.c_int(k),
                                   C.c_longlong(ignore_index), _ptr(confmat_u64), _ptr(confbin_u64), _ptr(counters_u64), _stream()),
           "la_confmat_update")


def resample_u8(inp, n_outer: int, in_size: int, inner: int, out_size: int, bounds_i32, kk_i32, out) -> None:
    _dev(inp)
    _check(lib().la_resample_u8(_ptr(inp), C.c_long(n_outer), C.c_int(in_size), C.c_int(inner), C.c_int(out_size), _ptr(bounds_i32),
                                _ptr(kk_i32), C.c_int(kk_i32.shape[1]), _ptr(out), _stream()), "la_resample_u8")


def prompt_masks(masks_u8, first_i32, count_i32, index_i32, p: int, h: int, w: int, nh: int, nw: int, s: int, mo: int, out, flags_u8) -> None:
    _dev(masks_u8)
    _check(lib().la_prompt_masks(_ptr(masks_u8), _ptr(first_i32), _ptr(count_i32), _ptr(index_i32), C.c_int(p), C.c_int(h), C.c_int(w),
                                 C.c_int(nh), C.c_int(nw), C.c_int(s), C.c_int(mo), _ptr(out), _ptr(flags_u8), _stream()), "la_prompt_masks")


def focal_loss(logits, target_i64, gamma: float, class_weighting: bool, scale: float, ignore_index: int, loss, dlogits, class_weights,
               scratch) -> None:
    _dev(logits)
    b, c = logits.shape[0], logits.shape[1]
    hw = logits.numel() // (b * c)
    _check(lib().la_focal_loss(_ptr(logits), _ptr(target_i64), C.c_int(b), C.c_int(c), C.c_long(hw), C.c_float(gamma), C.c_int(int(class_weighting)),
                               C.c_float(scale), C.c_longlong(ignore_index), _ptr(loss), _ptr(dlogits), _ptr(class_weights), _ptr(scratch),
                               C.c_long(scratch.numel() * scratch.element_size()), _stream()), "la_focal_loss")


def adamw_step(params, grads, exp_avg, exp_avg_sq, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               grad_scale: float = 1.0) -> None:
    _dev(params)
    for t in (params, grads, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != params.numel():
            raise ValueError("adamw_step: flat contiguous fp32 buffers of equal length expected")
    _check(lib().la_adamw_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), C.c_long(params.numel()), C.c_float(lr),
                               C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), C.c_int(step),
                               C.c_float(grad_scale), _stream()), "la_adamw_step")


def u8_to_chw_norm(inp, h: int, w: int, sh: int, sw: int, mean, std, out) -> None:
    _dev(inp)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    _check(lib().la_u8_to_chw_norm(_ptr(inp), C.c_int(h), C.c_int(w), C.c_int(sh), C.c_int(sw), m3, s3, _ptr(out), _stream()),
           "la_u8_to_chw_norm")


def conv3x3_f32(x32, b: int, h: int, w: int, cin: int, wt, bias, cout: int, out32) -> None:
    _dev(x32)
    _check(lib().la_conv3x3_f32(_ptr(x32), C.c_int(b), C.c_int(h), C.c_int(w), C.c_int(cin), _ptr(wt), _ptr(bias), C.c_int(cout),
                                _ptr(out32), _stream()), "la_conv3x3_f32")


def conv3x3_split_ok(cin: int, cout: int) -> bool:
    return bool(lib().la_conv3x3_split_ok(C.c_int(cin), C.c_int(cout)))


def conv3x3_split(x32, b: int, h: int, w: int, cin: int, wt, bias, cout: int, out32) -> None:
    """la_conv3x3_f32's convolution for 32 -> 32 channels as three fp16 MFMA products on plane pairs (fp32-class accuracy, ~3x the rate)."""
    _dev(x32)
    for t_, n_ in ((x32, b * h * w * cin), (wt, cout * 9 * cin), (out32, b * h * w * cout)):
        if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.numel() < n_:
            raise RuntimeError("conv3x3_split: contiguous fp32 input [b*h*w, cin], weight [cout, 9*cin], output [b*h*w, cout]")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() < cout):
        raise RuntimeError("conv3x3_split: bias must be fp32 [cout]")
    _check(lib().la_conv3x3_split(_ptr(x32), C.c_int(b), C.c_int(h), C.c_int(w), C.c_int(cin), _ptr(wt), _ptr(bias), C.c_int(cout),
                                  _ptr(out32), _stream()), "la_conv3x3_split")


# ---- backward kernels (training step, csrc/train.hip) ------------------------------------------------------------------------
def _f32c(*ts) -> None:
    for t in ts:
        if t is None:
            continue
        _dev(t)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("contiguous fp32 device tensors expected")


def _f32rows(t: torch.Tensor) -> None:
    _dev(t)
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("fp32 device matrices with unit column stride expected")


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None) -> None:
    """dw[N,K] += dy[M,N]^T @ x[M,K] (and db[N] += dy.sum(0) when db is given).  dy / x may be column slices of wider matrices (row
    stride = the parent's width)."""
    _f32rows(dy), _f32rows(x), _f32c(dw)
    m, n = dy.shape
    k = x.shape[1]
    if db is not None:
        _f32c(db)
        if db.numel() != n:
            raise ValueError("gemm_tn: db must hold N elements")
    _check(lib().la_gemm_tn_db(_ptr(dy), C.c_int(dy.stride(0)), _ptr(x), C.c_int(x.stride(0)), _ptr(dw), C.c_int(k), C.c_int(m), C.c_int(n),
                               C.c_int(k), _ptr(db), _stream()), "la_gemm_tn")


def gemm_tn16(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None, gsize: int = 0, gstride: int = 0) -> None:
    """dw[N,K] += dy[R,N]^T @ x[R,K] from row-major 16-bit operands (column slices allowed), db[N] += dy.sum(0); 16-bit MFMA, no
    transposed copies.  gsize / gstride: output row n -> dw row (n // gsize) * gstride + n % gsize (dw = the first row's tensor)."""
    _dev(dy)
    if dy.dtype not in (torch.float16, torch.bfloat16) or x.dtype != dy.dtype:
        raise TypeError("gemm_tn16: 16-bit operands of one dtype")
    if dy.stride(1) != 1 or x.stride(1) != 1 or dy.shape[0] != x.shape[0]:
        raise ValueError("gemm_tn16: row-major operands with the same number of rows")
    _f32c(dw)
    r, n = dy.shape
    k = x.shape[1]
    if db is not None:
        _f32c(db)
        if db.numel() != n:
            raise ValueError("gemm_tn16: db must hold N elements")
    _check(lib().la_gemm_tn16(_ptr(dy), C.c_int(dy.stride(0)), _ptr(x), C.c_int(x.stride(0)), _ptr(dw), C.c_int(k), C.c_int(r), C.c_int(n),
                              C.c_int(k), C.c_int(gsize), C.c_int(gstride), _ptr(db), C.c_int(dt_of(dy)), _stream()), "la_gemm_tn16")


def colsum_acc(dy: torch.Tensor, out: torch.Tensor) -> None:
    """out[N] += dy[M,N].sum(0) (dy may be a column slice of a wider matrix)."""
    _f32rows(dy), _f32c(out)
    m, n = dy.shape
    _check(lib().la_colsum_acc(_ptr(dy), C.c_int(dy.stride(0)), C.c_long(m), C.c_int(n), _ptr(out), _stream()), "la_colsum_acc")


def layernorm_bwd(x, dy, gamma, beta, eps: float, gelu: bool, dx, dgamma, dbeta) -> None:
    _f32c(x, dy, gamma, beta, dx, dgamma, dbeta)
    rows, e = x.shape
    _check(lib().la_layernorm_bwd(_ptr(x), _ptr(dy), C.c_long(rows), C.c_int(e), _ptr(gamma), _ptr(beta), C.c_float(eps), C.c_int(int(gelu)),
                                  _ptr(dx), _ptr(dgamma), _ptr(dbeta), _stream()), "la_layernorm_bwd")


def layernorm_bwd_res(x, dy, gamma, beta, eps: float, add, dx, out16, dgamma, dbeta) -> None:
    """dx = LayerNorm backward(x, dy) + add (add may be dx), optionally with a 16-bit copy of dx in out16."""
    _f32c(x, dy, gamma, beta, dx, dgamma, dbeta)
    if add is not None:
        _f32c(add)
    rows, e = x.shape
    _check(lib().la_layernorm_bwd_res(_ptr(x), _ptr(dy), C.c_long(rows), C.c_int(e), _ptr(gamma), _ptr(beta), C.c_float(eps), C.c_int(0),
                                      _ptr(add), _ptr(dx), _ptr(out16), C.c_int(dt_of(out16) if out16 is not None else LA_F16), _ptr(dgamma),
                                      _ptr(dbeta), _stream()), "la_layernorm_bwd_res")


def transpose_many(tile_table: torch.Tensor) -> None:
    """tile_table: int64 [ntiles, 4] on the device - (src, dst, rows << 32 | cols, r0 << 32 | c0) per 32 x 32 tile; dst[c][r] = src[r][c]."""
    _dev(tile_table)
    if tile_table.dtype != torch.int64 or tile_table.dim() != 2 or tile_table.shape[1] != 4 or not tile_table.is_contiguous():
        raise ValueError("transpose_many needs a contiguous int64 [ntiles, 4] table")
    _check(lib().la_transpose_many(_ptr(tile_table), C.c_int(tile_table.shape[0]), _stream()), "la_transpose_many")


def act_fwd(x, y, kind: int) -> None:
    _f32c(x, y)
    _check(lib().la_act_fwd(_ptr(x), _ptr(y), C.c_long(x.numel()), C.c_int(kind), _stream()), "la_act_fwd")


def act_bwd(x, dy, dx, kind: int) -> None:
    _f32c(x, dy, dx)
    _check(lib().la_act_bwd(_ptr(x), _ptr(dy), _ptr(dx), C.c_long(x.numel()), C.c_int(kind), _stream()), "la_act_bwd")


def attn_small_lse(q, k, b: int, nq: int, nk: int, heads: int, hd: int, lse) -> None:
    _check(lib().la_attn_small_lse(_ptr(q), C.c_int(q.stride(0)), _ptr(k), C.c_int(k.stride(0)), C.c_int(b), C.c_int(nq), C.c_int(nk),
                                   C.c_int(heads), C.c_int(hd), _ptr(lse), _stream()), "la_attn_small_lse")


def attn_small_bwd(q, k, v, o, dout, lse, b: int, nq: int, nk: int, heads: int, hd: int, dq, dk, dv) -> None:
    _check(lib().la_attn_small_bwd(_ptr(q), C.c_int(q.stride(0)), _ptr(k), C.c_int(k.stride(0)), _ptr(v), C.c_int(v.stride(0)), _ptr(o),
                                   _ptr(dout), C.c_int(dout.stride(0)), _ptr(lse), C.c_int(b), C.c_int(nq), C.c_int(nk), C.c_int(heads),
                                   C.c_int(hd), _ptr(dq), _ptr(dk), _ptr(dv), _stream()), "la_attn_small_bwd")


def bilinear_rows(x, n: int, h: int, w: int, c: int, oh: int, ow: int, out) -> None:
    """la_bilinear_rows: NHWC rows [n, h * w, c] -> [n, oh * ow, c] (bilinear, align_corners=False)."""
    _f32c(x, out)
    if c % 4 or x.numel() != n * h * w * c or out.numel() != n * oh * ow * c:
        raise ValueError(f"la_bilinear_rows: [{n}, {h}x{w}, {c}] -> [{n}, {oh}x{ow}, {c}] needs c % 4 == 0 and tensors of exactly those "
                         f"sizes (got {x.numel()} and {out.numel()} elements)")
    _check(lib().la_bilinear_rows(_ptr(x), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(c), _ptr(out), C.c_int(oh), C.c_int(ow), _stream()), "la_bilinear_rows")


def bilinear_rows_bwd_set(dy, n: int, oh: int, ow: int, c: int, dx, ih: int, iw: int) -> None:
    """la_bilinear_rows_bwd_set: adjoint of bilinear_rows for reductions (bilinear_bwd_set_ok), dx written."""
    _f32c(dy, dx)
    if c % 4 or dy.numel() != n * oh * ow * c or dx.numel() != n * ih * iw * c or not bilinear_bwd_set_ok(oh, ow, ih, iw):
        raise ValueError(f"la_bilinear_rows_bwd_set: dy [{n}, {oh}x{ow}, {c}] -> dx [{n}, {ih}x{iw}, {c}] needs c % 4 == 0, a reduction "
                         f"the gathered adjoint takes (bilinear_bwd_set_ok) and tensors of exactly those sizes (got {dy.numel()} and {dx.numel()})")
    _check(lib().la_bilinear_rows_bwd_set(_ptr(dy), C.c_int(n), C.c_int(oh), C.c_int(ow), C.c_int(c), _ptr(dx), C.c_int(ih), C.c_int(iw), _stream()),
           "la_bilinear_rows_bwd_set")


def bilinear_bwd_set_ok(oh: int, ow: int, ih: int, iw: int) -> bool:
    """True when la_bilinear_bwd_set takes the shape (a reduction that fits its LDS tables): dx is then WRITTEN, not accumulated."""
    return bool(lib().la_bilinear_bwd_set_ok(C.c_int(oh), C.c_int(ow), C.c_int(ih), C.c_int(iw)))


def bilinear_bwd_set(dy, n: int, oh: int, ow: int, dy_plane: int, dy_ld: int, dx, ih: int, iw: int, dx_plane: int, dx_ld: int) -> None:
    """la_bilinear_bwd_set: the adjoint of a bilinear REDUCTION as a gather (no atomics, dx written)."""
    _f32c(dy, dx)
    if n <= 0 or dy_ld < ow or dx_ld < iw or (n - 1) * dy_plane + (oh - 1) * dy_ld + ow > dy.numel() or \
            (n - 1) * dx_plane + (ih - 1) * dx_ld + iw > dx.numel() or not bilinear_bwd_set_ok(oh, ow, ih, iw):
        raise ValueError(f"la_bilinear_bwd_set: {n} planes {oh}x{ow} (plane {dy_plane}, ld {dy_ld}) -> {ih}x{iw} (plane {dx_plane}, ld {dx_ld}) "
                         f"reaches beyond the tensors ({dy.numel()} / {dx.numel()} elements) or is not a reduction the gathered adjoint takes")
    _check(lib().la_bilinear_bwd_set(_ptr(dy), C.c_int(n), C.c_int(oh), C.c_int(ow), C.c_long(dy_plane), C.c_int(dy_ld), _ptr(dx), C.c_int(ih),
                                     C.c_int(iw), C.c_long(dx_plane), C.c_int(dx_ld), _stream()), "la_bilinear_bwd_set")


def bilinear_bwd(dy, n: int, oh: int, ow: int, dy_plane: int, dy_ld: int, dx, ih: int, iw: int, dx_plane: int, dx_ld: int) -> None:
    _dev(dy)
    _check(lib().la_bilinear_bwd(_ptr(dy), C.c_int(n), C.c_int(oh), C.c_int(ow), C.c_long(dy_plane), C.c_int(dy_ld), _ptr(dx), C.c_int(ih),
                                 C.c_int(iw), C.c_long(dx_plane), C.c_int(dx_ld), _stream()), "la_bilinear_bwd")


def classify_bwd(dseg, feat, protos, b: int, npix: int, c: int, cf: int, dfeat, dprotos) -> None:
    _f32c(dseg, feat, protos, dfeat, dprotos)
    _check(lib().la_classify_bwd(_ptr(dseg), _ptr(feat), _ptr(protos), C.c_int(b), C.c_int(npix), C.c_int(c), C.c_int(cf), _ptr(dfeat),
                                 _ptr(dprotos), _stream()), "la_classify_bwd")


def row_broadcast(src, groups: int, rep: int, d: int, scale: float, out) -> None:
    _f32c(src, out)
    _check(lib().la_row_broadcast(_ptr(src), C.c_long(groups), C.c_int(rep), C.c_int(d), C.c_float(scale), _ptr(out), _stream()),
           "la_row_broadcast")


# ---- fused image-side kernels of the two-way transformer (csrc/twoway.hip) -----------------------------------------------------
def twoway_part_size(groups: int, hw: int, nt: int, d: int) -> int:
    """fp32 elements of la_twoway_t2i's partials buffer: [groups][64-row tiles][2 row tiles of 32][nt][8 heads][2 + head width]."""
    return groups * ((hw + 63) // 64) * 2 * nt * 8 * (2 + d // 16)


def twoway_pe_layout(table: torch.Tensor) -> torch.Tensor:
    """pe @ W.T + b as fp32 [hw, D / 2] -> the same table in the order the fused two-way kernels read it (one pass, once per layer and grid)."""
    _f32c(table)
    hw, di = table.shape
    out = torch.empty(((hw + 63) // 64) * 64 * di, device=table.device)
    _check(lib().la_twoway_pe_layout(_ptr(table), C.c_int(hw), C.c_int(di), _ptr(out), _stream()), "la_twoway_pe_layout")
    return out


def twoway_t2i(img, wk, wv, pek, bv, q, groups: int, hw: int, nt: int, heads: int, part, out) -> None:
    """wk / wv: (hi, lo) fp16 plane pairs [D / 2, D]; pek = twoway_pe_layout(pe @ Wk.T + bk).  D = 256 or 512, 8 heads."""
    _f32c(img, pek, bv, q, part, out)
    d = img.shape[1]
    need = twoway_part_size(groups, hw, nt, d)
    if part.numel() < need:
        raise ValueError(f"twoway_t2i: scratch too small ({part.numel()} < {need})")
    _check(lib().la_twoway_t2i(_ptr(img), _ptr(wk[0]), _ptr(wk[1]), _ptr(wv[0]), _ptr(wv[1]), _ptr(pek), _ptr(bv), _ptr(q),
                               C.c_int(groups), C.c_int(hw), C.c_int(nt), C.c_int(d), C.c_int(heads), _ptr(part), _ptr(out), _stream()),
           "la_twoway_t2i")


def twoway_i2t(img, wq, peq, k, v, wo, bo, gamma, beta, eps: float, groups: int, hw: int, nt: int, heads: int) -> None:
    """peq = twoway_pe_layout(pe @ Wq.T + bq); img is updated in place."""
    _f32c(img, peq, k, v, bo, gamma, beta)
    _check(lib().la_twoway_i2t(_ptr(img), _ptr(wq[0]), _ptr(wq[1]), _ptr(peq), _ptr(k), _ptr(v), _ptr(wo[0]), _ptr(wo[1]), _ptr(bo),
                               _ptr(gamma), _ptr(beta), C.c_float(eps), C.c_int(groups), C.c_int(hw), C.c_int(nt), C.c_int(img.shape[1]),
                               C.c_int(heads), _stream()), "la_twoway_i2t")


# ---- image-encoder backward ---------------------------------------------------------------------------
def attn_fwd_lse(qkv, vt, out16, lse, b: int, heads: int, t: int, tpad: int, e: int, scale: float) -> None:
    _dev(qkv)
    _check(lib().la_attn_fwd_lse(_ptr(qkv), _ptr(vt), _ptr(out16), _ptr(lse), C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad),
                                 C.c_int(e), C.c_float(scale), C.c_int(dt_of(qkv)), _stream()), "la_attn_fwd_lse")


def head_transpose(src, col0: int, b: int, heads: int, t: int, tpad: int, dst) -> None:
    _dev(src)
    _check(lib().la_head_transpose(_ptr(src), C.c_int(src.stride(0)), C.c_int(col0), C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad),
                                   _ptr(dst), C.c_int(dt_of(src)), _stream()), "la_head_transpose")


def attn_bwd(qkv, out16, dout16, kt, qt, dot, lse, dvec, dqkv, b: int, heads: int, t: int, tpad: int, e: int, scale: float) -> None:
    _dev(qkv)
    _check(lib().la_attn_bwd(_ptr(qkv), _ptr(out16), _ptr(dout16), _ptr(kt), _ptr(qt), _ptr(dot), _ptr(lse), _ptr(dvec), _ptr(dqkv),
                             C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad), C.c_int(e), C.c_float(scale), C.c_int(dt_of(qkv)),
                             _stream()), "la_attn_bwd")


def attn_fwd_relpos_lse(qkv, vt, out16, relh, relw, lse, b: int, heads: int, t: int, tpad: int, g: int, e: int, scale: float) -> None:
    _dev(qkv)
    _check(lib().la_attn_fwd_relpos_lse(_ptr(qkv), _ptr(vt), _ptr(out16), _ptr(relh), _ptr(relw), _ptr(lse), C.c_int(b), C.c_int(heads),
                                        C.c_int(t), C.c_int(tpad), C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(dt_of(qkv)), _stream()),
           "la_attn_fwd_relpos_lse")


def attn_bwd_relpos(qkv, out16, dout16, kt, qt, dot, lse, dvec, dqkv, relh, relw, drelh, drelw, b: int, heads: int, t: int, tpad: int, g: int,
                    e: int, scale: float) -> None:
    _dev(qkv)
    _check(lib().la_attn_bwd_relpos(_ptr(qkv), _ptr(out16), _ptr(dout16), _ptr(kt), _ptr(qt), _ptr(dot), _ptr(lse), _ptr(dvec), _ptr(dqkv),
                                    _ptr(relh), _ptr(relw), _ptr(drelh), _ptr(drelw), C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad),
                                    C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(dt_of(qkv)), _stream()), "la_attn_bwd_relpos")


def relpos_bwd(qkv, dqkv, drelh, drelw, tabh, tabw, dtabh, dtabw, b: int, heads: int, g: int, e: int, gscale: float = 1.0) -> None:
    _dev(qkv)
    _check(lib().la_relpos_bwd(_ptr(qkv), _ptr(dqkv), _ptr(drelh), _ptr(drelw), _ptr(tabh), _ptr(tabw), _ptr(dtabh), _ptr(dtabw), C.c_int(b),
                               C.c_int(heads), C.c_int(g), C.c_int(e), C.c_float(gscale), C.c_int(dt_of(qkv)), _stream()), "la_relpos_bwd")


def cast(src, dst, scale: float = 1.0) -> None:
    """dst = scale * src (fp32 <-> 16-bit, or fp32 -> fp32 possibly in place); contiguous tensors of equal numel."""
    _dev(src)
    if not (src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()):
        raise ValueError("la_cast needs contiguous tensors of equal size")
    _check(lib().la_cast(_ptr(src), C.c_int(dt_of(src)), _ptr(dst), C.c_int(dt_of(dst)), C.c_long(src.numel()), C.c_float(scale), _stream()),
           "la_cast")


def gelu_fwd16(pre16, post16) -> None:
    """post16 = GELU(pre16), contiguous 16-bit tensors of equal size (numel % 8 == 0)."""
    _dev(pre16)
    if not (pre16.is_contiguous() and post16.is_contiguous() and pre16.numel() == post16.numel() and pre16.dtype == post16.dtype):
        raise ValueError("gelu_fwd16 needs contiguous 16-bit tensors of equal size and dtype")
    _check(lib().la_gelu_fwd16(_ptr(pre16), _ptr(post16), C.c_long(pre16.numel()), C.c_int(dt_of(pre16)), _stream()), "la_gelu_fwd16")


def gelu_bwd16(pre16, dh, d32=None, d16=None) -> None:
    _dev(pre16)
    _check(lib().la_gelu_bwd16(_ptr(pre16), _ptr(dh), _ptr(d32), _ptr(d16), C.c_long(pre16.numel()), C.c_int(dt_of(pre16)), _stream()),
           "la_gelu_bwd16")


def axpy(x, y, a: float) -> None:
    """y += a * x (contiguous fp32)."""
    _f32c(x, y)
    _check(lib().la_axpy(_ptr(x), _ptr(y), C.c_long(x.numel()), C.c_float(a), _stream()), "la_axpy")


def transpose16(src, dst, colsum=None) -> None:
    """dst[c, r] = src[r, c]; src fp32 / 16-bit [R, C] (row stride src.stride(0)), dst 16-bit [C, Rp] with Rp >= R, Rp % 64 == 0 (zero padded).
    colsum: optional fp32 [C] that receives += the column sums of the (16-bit) values written - the bias gradient of the layer whose
    output gradient is being transposed."""
    _dev(src)
    r, c = src.shape
    if src.stride(1) != 1 or not dst.is_contiguous() or dst.shape[0] != c:
        raise ValueError("transpose16: src needs unit column stride, dst contiguous [C, Rp]")
    if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != c or not colsum.is_contiguous()):
        raise ValueError("transpose16: colsum must be contiguous fp32 [C]")
    _check(lib().la_transpose16(_ptr(src), C.c_int(dt_of(src)), C.c_int(src.stride(0)), C.c_int(r), C.c_int(c), _ptr(dst), C.c_int(dt_of(dst)),
                                C.c_int(dst.shape[1]), _ptr(colsum), _stream()), "la_transpose16")


# ---- fp8 QK^T attention (opt-in) --------------------------------------------------------------------------
def qk_fp8(qkv, e: int, qk8) -> None:
    _dev(qkv)
    _check(lib().la_qk_fp8(_ptr(qkv), C.c_long(qkv.shape[0]), C.c_int(e), _ptr(qk8), C.c_int(dt_of(qkv)), _stream()), "la_qk_fp8")


def add_rowvec_split(x, v, rows_per_group: int, split16) -> None:
    """x[r] += v[r / rows_per_group] in place (v may be None) and split16 [rows, 2 D] fp16 = [hi | lo] planes of the result."""
    _f32c(x)
    if v is not None:
        _f32c(v)
    if split16.dtype != torch.float16 or tuple(split16.shape) != (x.shape[0], 2 * x.shape[1]) or not split16.is_contiguous():
        raise ValueError("add_rowvec_split: split16 must be a contiguous fp16 [rows, 2 D] buffer")
    _check(lib().la_add_rowvec_split(_ptr(x), _ptr(v), C.c_long(x.shape[0]), C.c_int(rows_per_group), C.c_int(x.shape[1]), _ptr(split16),
                                     _stream()), "la_add_rowvec_split")


def attn_fwd_fp8(qk8, vt, out16, b: int, heads: int, t: int, tpad: int, e: int, scale: float) -> None:
    _dev(qk8)
    _check(lib().la_attn_fwd_fp8(_ptr(qk8), _ptr(vt), _ptr(out16), C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad), C.c_int(e),
                                 C.c_float(scale), C.c_int(dt_of(out16)), _stream()), "la_attn_fwd_fp8")


# ---- token-mean correction of single-plane weights ---------------------------------------------------------
def colmean16(src, groups: int, rows_per_group: int, out, scratch, wpart: int = 0, h: int = 0, w: int = 0) -> None:
    """out[g] = mean of the rows of group g (fp32 [groups, D]); scratch fp32 >= groups * ceil(rows_per_group / 128) * D elements;
    wpart: src window-partitioned, rows gathered in image order."""
    _dev(src)
    need = groups * ((rows_per_group + 127) // 128) * src.shape[1]
    if scratch.numel() < need or scratch.dtype != torch.float32:
        raise ValueError(f"colmean16 scratch needs {need} fp32 elements")
    _check(lib().la_colmean16(_ptr(src), C.c_int(src.stride(0)), C.c_int(groups), C.c_int(rows_per_group), C.c_int(src.shape[1]), _ptr(out),
                              _ptr(scratch), C.c_int(wpart), C.c_int(h), C.c_int(w), C.c_int(dt_of(src)), _stream()), "la_colmean16")


LN_CS_ROWS = 32        # rows per column-sum partial of la_layernorm_g (norm.hip)


def ln_cs_chunks(rows_per_group: int) -> int:
    return (rows_per_group + LN_CS_ROWS - 1) // LN_CS_ROWS


def layernorm_g(x, xg, rows_per_group: int, gamma, beta, eps: float, *, out32=None, out16=None, window=0, H=0, W=0, colsum_part=None,
                dt=LA_F16) -> None:
    """colsum_part: fp32, >= (rows / rows_per_group) * ln_cs_chunks(rows_per_group) * E elements (column sums of the stored rows per
    fixed share of a group; colsum_fold makes the means)."""
    _dev(x)
    rows, e = x.shape
    if colsum_part is not None:
        need = rows // rows_per_group * ln_cs_chunks(rows_per_group) * e
        if colsum_part.dtype != torch.float32 or colsum_part.numel() < need:
            raise ValueError(f"layernorm_g colsum_part needs {need} fp32 elements")
    _check(lib().la_layernorm_g(_ptr(x), _ptr(xg), C.c_int(rows_per_group), C.c_int(x.stride(0)), C.c_int(rows), C.c_int(e), _ptr(gamma),
                                _ptr(beta), C.c_float(eps), _ptr(out32), _ptr(out16), C.c_int(window), C.c_int(H), C.c_int(W),
                                _ptr(colsum_part), C.c_int(dt), _stream()), "la_layernorm_g")


def attn_fwd_cs(qkv, vt, out16, relh, relw, b: int, heads: int, t: int, tpad: int, g: int, e: int, scale: float, mode: int, cspart,
                cs_h: int = 0, cs_w: int = 0, tabh=None, tabw=None) -> None:
    """attn_fwd + column sums of every 128-query block of the output: cspart fp32 [b * ceil(t / 128), e]."""
    need = b * ((t + 127) // 128) * e
    if cspart.dtype != torch.float32 or cspart.numel() < need:
        raise ValueError(f"attn_fwd_cs cspart needs {need} fp32 elements")
    _check(lib().la_attn_fwd_cs(_ptr(qkv), _ptr(vt), _ptr(out16), _ptr(relh), _ptr(relw), _ptr(tabh), _ptr(tabw), C.c_int(b), C.c_int(heads),
                                C.c_int(t), C.c_int(tpad), C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(mode), _ptr(cspart),
                                C.c_int(cs_h), C.c_int(cs_w), C.c_int(dt_of(qkv)), _stream()), "la_attn_fwd_cs")


def attn_fwd_rows(qkv, out16, b: int, heads: int, t: int, tpad: int, g: int, e: int, scale: float, mode: int, tabh=None, tabw=None,
                  cspart=None, img_hw=None, padrow=None) -> None:
    """Attention without a V^T copy (la_attn_fwd_rows): V tiles row-major from the v columns of qkv.  img_hw = (H, W) + padrow: SAM windows
    addressed in image order (b = images * windows per image)."""
    _dev(qkv)
    if qkv.dtype not in (torch.float16, torch.bfloat16) or out16.dtype != qkv.dtype or not (qkv.is_contiguous() and out16.is_contiguous()):
        raise ValueError("attn_fwd_rows: qkv / out16 must be contiguous 16-bit tensors of one dtype")
    if cspart is not None:
        need = b * ((t + 127) // 128) * e
        if cspart.dtype != torch.float32 or cspart.numel() < need:
            raise ValueError(f"attn_fwd_rows cspart needs {need} fp32 elements")
    ih, iw = (0, 0) if img_hw is None else img_hw
    if padrow is not None and (padrow.dtype != qkv.dtype or padrow.numel() != 3 * e or not padrow.is_contiguous()):
        raise ValueError("attn_fwd_rows: padrow must be a contiguous [3E] row of the qkv dtype")
    _check(lib().la_attn_fwd_rows(_ptr(qkv), _ptr(out16), _ptr(tabh), _ptr(tabw), C.c_int(b), C.c_int(heads), C.c_int(t), C.c_int(tpad),
                                  C.c_int(g), C.c_int(e), C.c_float(scale), C.c_int(mode), _ptr(cspart), C.c_int(ih), C.c_int(iw), _ptr(padrow),
                                  C.c_int(dt_of(qkv)), _stream()), "la_attn_fwd_rows")


def colsum_fold(part, groups: int, chunks: int, d: int, inv: float, out) -> None:
    """out[g, :d] = inv * sum_j part[g * chunks + j, :d] (out: fp32 rows of stride out.stride(0), e.g. a column slice)."""
    _dev(part)
    if out.dtype != torch.float32 or out.stride(1) != 1 or out.shape[0] < groups or out.shape[1] < d:
        raise ValueError("colsum_fold: out must be fp32 [groups, >= d] with unit column stride")
    _check(lib().la_colsum_fold(_ptr(part), C.c_int(groups), C.c_int(chunks), C.c_int(d), C.c_float(inv), _ptr(out), C.c_int(out.stride(0)),
                                _stream()), "la_colsum_fold")


def add_rowvec(x, v, rows_per_group: int) -> None:
    _f32c(x, v)
    _check(lib().la_add_rowvec(_ptr(x), _ptr(v), C.c_long(x.shape[0]), C.c_int(rows_per_group), C.c_int(x.shape[1]), _stream()), "la_add_rowvec")
