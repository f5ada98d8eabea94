"""Differentiable decoder ops: ``torch.autograd.Function`` shells whose forward AND backward are HIP kernels of libla_hip.so.

Training side of the hot path.  torch.autograd only records the graph and moves data (views, permutes, cat,
index_select, expand); every arithmetic node of the prompt encoder / mask decoder / necks is one of the functions below:

    linear        la_gemm (exact-fp32 MFMA)            bwd: la_gemm (dX), la_gemm_tn (dW), la_colsum_acc (db)
    layer_norm    la_layernorm                         bwd: la_layernorm_bwd           (nn.LayerNorm, LayerNorm2d in NHWC, +GELU)
    act           la_act_fwd                           bwd: la_act_bwd                 (GELU erf, ReLU)
    attention     la_attn_small                        bwd: la_attn_small_lse + la_attn_small_bwd
    add_rows      la_add_cast (row-periodic operand)   bwd: identity / la_gemm_tn-free row fold via torch view-sum
    mean_rows     la_colmean                           bwd: la_row_broadcast
    conv3x3       la_conv3x3_f32 (implicit GEMM)       bwd: la_conv3x3_f32 with the transposed taps (dX), la_im2col_3x3 + la_gemm_tn (dW)
    classify      la_classify                          bwd: la_classify_bwd
    bilinear      la_bilinear                          bwd: la_bilinear_bwd

All tensors are fp32, contiguous, on the device; 2-D activations are [rows, channels] (NHWC rows), as in the inference engine.
Reference graph: label_anything/models/{common,transformer,prompt_encoder,mask_decoder}.py under experiment/utils.py:266-303.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch.autograd import Function

from . import _lib as L

Tensor = torch.Tensor

def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


class WeightTransposes:
    """W^T copies for the data gradients dX = dY W of the nn.Linear layers, refreshed by ONE launch per training step.

    A trainer whose weights keep their addresses (views of one flat parameter buffer) owns an enabled instance, installs it as
    ``autograd_ops.WT`` and calls ``invalidate()`` at the start of every ``forward_backward`` (the weights may have changed); the first
    request after that re-transposes every matrix seen so far with one ``la_transpose_many`` launch.  A matrix seen for the first time
    is transposed on its own and joins the table.  The module's default instance is off: every backward transposes its own weight."""

    def __init__(self, enabled: bool = False, lo: int = 0, hi: int = 0) -> None:
        self.enabled = enabled
        self.lo, self.hi = lo, hi    # address range of the flat parameter buffer: only matrices inside it keep their address from step to
        self.entries = {}            # step (a reshaped / permuted weight is a fresh temporary every time); (data_ptr, rows, cols) -> W^T
        self.table = None
        self.fresh = False

    def invalidate(self) -> None:
        self.fresh = False

    def get(self, w: Tensor) -> Tensor:
        if not self.enabled or not (self.lo <= w.data_ptr() < self.hi):
            wt = w.new_empty(w.shape[1], w.shape[0])
            L.nhwc_to_nchw(w, 1, w.shape[1], w.shape[0], wt)
            return wt
        key = (w.data_ptr(), w.shape[0], w.shape[1])
        wt = self.entries.get(key)
        if wt is None:
            wt = w.new_empty(w.shape[1], w.shape[0])
            L.nhwc_to_nchw(w, 1, w.shape[1], w.shape[0], wt)
            self.entries[key] = wt
            self.table = None
            return wt
        if not self.fresh:
            if self.table is None:
                rows = []
                for (ptr, r, c), t in self.entries.items():
                    for r0 in range(0, r, 32):
                        for c0 in range(0, c, 32):
                            rows.append((ptr, t.data_ptr(), (r << 32) | c, (r0 << 32) | c0))
                self.table = torch.tensor(rows, dtype=torch.int64).to(w.device)
            L.transpose_many(self.table)
            self.fresh = True
        return wt


WT = WeightTransposes()


class GradSink:
    """Where the parameter gradients of the kernels below are ADDED directly: a trainer whose parameters' ``.grad`` are views of one flat
    buffer installs its sink as ``autograd_ops.SINK`` for the duration of a backward pass.  la_gemm_tn / la_layernorm_bwd accumulate
    into their output anyway, so a backward node that finds its parameter here adds into the flat buffer and returns no gradient for it -
    instead of zero-filling a temporary for the kernel and leaving autograd to add that temporary to ``.grad`` (four tiny launches per
    layer: ~340 per decoder step, 3 ms of its 40).  ``touch(i)`` stands in for the post-accumulate hook of parameter i.
    The module's default sink is empty: every node returns its gradients to autograd."""

    def __init__(self, params=(), grad_views=(), touch=None) -> None:
        self.map = {p.data_ptr(): (i, g) for i, (p, g) in enumerate(zip(params, grad_views)) if p.numel() > 0}
        self.numel = {p.data_ptr(): p.numel() for p in params}
        self.touch = touch

    def find(self, t):
        """The flat-buffer gradient of parameter tensor t (or a contiguous reshape of it), shaped like t; None if t is something else."""
        if t is None or not self.map or not t.is_contiguous():
            return None
        return self.find_ptr(t.data_ptr(), tuple(t.shape))

    def find_ptr(self, ptr: int, shape):
        hit = self.map.get(ptr)
        if hit is None or self.numel[ptr] != math.prod(shape):
            return None
        i, g = hit
        self.touch(i)
        return g.view(shape)


SINK = GradSink()


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _c(x), _c(w)
        y = x.new_empty(x.shape[0], w.shape[0])
        L.gemm(x, w, bias=b, out32=y)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bias_key = (b.data_ptr(), tuple(b.shape)) if b is not None and b.is_contiguous() else None      # (for the gradient sink)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wt = WT.get(w)                                   # W^T in nn.Linear layout (torch's strided copy of 85 weights was 3 ms of
            L.gemm(dy, wt, out32=dx)                         # the training step; one batched launch per step under a trainer): dY . W
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        gw = SINK.find(w) if ctx.needs_input_grad[1] else None
        gb = SINK.find_ptr(*ctx.bias_key) if want_db and ctx.bias_key is not None and SINK.map else None
        if want_db and gb is None:
            db = dy.new_zeros(dy.shape[1])
        bsum = gb if gb is not None else db
        if ctx.needs_input_grad[1]:
            if gw is None:
                dw = torch.zeros_like(w)
            L.gemm_tn(dy, x, gw if gw is not None else dw, bsum if want_db else None)   # dY^T . X (+ the column sums of dY, same pass)
        elif want_db:
            L.colsum_acc(dy, bsum)
        return dx, dw, db


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return _Linear.apply(x, w, b)


class _LinearFan(Function):
    """y_i = (x [+ pe]) W_i^T + b_i for i < n: several projections of ONE input as one autograd node - the image-side stream of the
    TwoWayTransformer feeds k_proj / v_proj of tokens -> image and q_proj of image -> tokens in the same layer (transformer.py:255-329,
    common.py:103-105), k and q from keys + pe.  Forward: the same launches as n separate ``linear`` calls (bit-identical outputs); backward:
    the n data gradients are added in ONE buffer through the GEMM's fp32 residual epilogue instead of n - 1 fan-in adds by autograd (three
    passes over the 270000 x 256 stream each on cfg3), and keys + pe is formed once.  pe: [period, D] without gradient, or None."""

    @staticmethod
    def forward(ctx, x, pe, use_pe, *wb):
        x = _c(x)
        n = len(wb) // 2
        xp = x
        if pe is not None and any(use_pe):
            if pe.requires_grad:
                raise ValueError("linear_fan: pe must not require a gradient (the node returns none for it; use add_rows + linear for a trainable pe)")
            if pe.dim() != 2 or pe.shape[1] != x.shape[1] or pe.shape[0] == 0 or x.shape[0] % pe.shape[0]:
                raise ValueError(f"linear_fan: pe {tuple(pe.shape)} must be [period, {x.shape[1]}] with period dividing the {x.shape[0]} rows of x")
            pe = _c(pe)
            xp = torch.empty_like(x)
            L.add_cast(x, pe, pe.shape[0] if pe.shape[0] != x.shape[0] else 0, out32=xp, dt=L.LA_F32)
        ws, outs, bkeys = [], [], []
        for i in range(n):
            w, b = _c(wb[2 * i]), wb[2 * i + 1]
            y = x.new_empty(x.shape[0], w.shape[0])
            L.gemm(xp if use_pe[i] else x, w, bias=b, out32=y)
            ws.append(w)
            outs.append(y)
            bkeys.append((b.data_ptr(), tuple(b.shape)) if b is not None and b.is_contiguous() else None)
        ctx.save_for_backward(x, xp, *ws)
        ctx.use_pe, ctx.has_bias, ctx.bias_keys, ctx.n = tuple(use_pe), [wb[2 * i + 1] is not None for i in range(n)], bkeys, n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, xp, *ws = ctx.saved_tensors
        n = ctx.n
        dx = None
        grads = [None] * (2 * n)
        for i in range(n):
            if dys[i] is None:
                continue
            dy, w = _c(dys[i]), ws[i]
            if ctx.needs_input_grad[0]:
                wt = WT.get(w)
                if dx is None:
                    dx = torch.empty_like(x)
                    L.gemm(dy, wt, out32=dx)
                else:
                    L.gemm(dy, wt, res=dx, out32=dx)            # dx += dY_i . W_i in the epilogue
            need_w, need_b = ctx.needs_input_grad[3 + 2 * i], ctx.has_bias[i] and ctx.needs_input_grad[4 + 2 * i]
            gw = SINK.find(w) if need_w else None
            gb = SINK.find_ptr(*ctx.bias_keys[i]) if need_b and ctx.bias_keys[i] is not None and SINK.map else None
            if need_b and gb is None:
                grads[2 * i + 1] = dy.new_zeros(dy.shape[1])
            bsum = gb if gb is not None else grads[2 * i + 1]
            if need_w:
                if gw is None:
                    grads[2 * i] = torch.zeros_like(w)
                L.gemm_tn(dy, xp if ctx.use_pe[i] else x, gw if gw is not None else grads[2 * i], bsum if need_b else None)
            elif need_b:
                L.colsum_acc(dy, bsum)
        return (dx, None, None, *grads)


def linear_fan(x: Tensor, pe: Optional[Tensor], projections) -> tuple:
    """projections: [(weight, bias or None, add_pe), ...] -> one output per entry (see _LinearFan)."""
    flat = []
    for w, b, _ in projections:
        flat += [w, b]
    return _LinearFan.apply(x, pe, tuple(bool(f) for _, _, f in projections), *flat)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, gelu):
        x = _c(x)
        y = torch.empty_like(x)
        L.layernorm(x, gamma, beta, eps, gelu=gelu, out32=y, dt=L.LA_F32)
        ctx.save_for_backward(x, gamma, beta)
        ctx.eps, ctx.gelu = eps, gelu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        dx = torch.empty_like(x)
        sg, sb = SINK.find(gamma), SINK.find(beta)
        if sg is not None and sb is not None:                # (both or neither: one kernel writes the pair)
            L.layernorm_bwd(x, _c(dy), gamma, beta, ctx.eps, ctx.gelu, dx, sg, sb)
            return dx, None, None, None, None
        dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
        L.layernorm_bwd(x, _c(dy), _c(gamma), _c(beta), ctx.eps, ctx.gelu, dx, dg, db)
        return dx, dg, db, None, None


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, gelu: bool = False) -> Tensor:
    return _LayerNorm.apply(x, gamma, beta, float(eps), bool(gelu))


class _Act(Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _c(x)
        y = torch.empty_like(x)
        L.act_fwd(x, y, kind)
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.act_bwd(x, _c(dy), dx, ctx.kind)
        return dx, None


def gelu(x: Tensor) -> Tensor:
    return _Act.apply(x, L.ACT_GELU)


def relu(x: Tensor) -> Tensor:
    return _Act.apply(x, L.ACT_RELU)


class _Attention(Function):
    """softmax(q k^T / sqrt(hd)) v per (group, head).  q [groups*nq, heads*hd], k / v [groups*nk, heads*hd]."""

    @staticmethod
    def forward(ctx, q, k, v, groups, nq, nk, heads):
        q, k, v = _c(q), _c(k), _c(v)
        hd = q.shape[1] // heads
        o = torch.empty_like(q)
        L.attn_small(q, k, v, groups, nq, nk, heads, hd, out32=o, dt=L.LA_F32)
        ctx.save_for_backward(q, k, v, o)
        ctx.dims = (groups, nq, nk, heads, hd)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        groups, nq, nk, heads, hd = ctx.dims
        do = _c(do)
        fewkeys = nk <= 256 and (nk <= nq or nq > 256)
        if fewkeys:
            dq, dk, dv = torch.empty_like(q), torch.zeros_like(k), torch.zeros_like(v)
            L.attn_small_bwd(q, k, v, o, do, None, groups, nq, nk, heads, hd, dq, dk, dv)
        else:
            lse = q.new_empty(groups * nq * heads)
            L.attn_small_lse(q, k, groups, nq, nk, heads, hd, lse)
            dq, dk, dv = torch.zeros_like(q), torch.empty_like(k), torch.empty_like(v)
            L.attn_small_bwd(q, k, v, o, do, lse, groups, nq, nk, heads, hd, dq, dk, dv)
        return dq, dk, dv, None, None, None, None


def attention(q: Tensor, k: Tensor, v: Tensor, groups: int, nq: int, nk: int, heads: int) -> Tensor:
    return _Attention.apply(q, k, v, groups, nq, nk, heads)


class _AddRows(Function):
    """x [rows, D] + y [ymod, D] repeated with period ymod (ymod == rows: plain add)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        out = torch.empty_like(x)
        L.add_cast(x, y, y.shape[0] if y.shape[0] != x.shape[0] else 0, out32=out, dt=L.LA_F32)
        ctx.ymod, ctx.rows = y.shape[0], x.shape[0]
        return out

    @staticmethod
    def backward(ctx, d):
        dy = None
        if ctx.needs_input_grad[1]:
            dy = d if ctx.ymod == ctx.rows else d.reshape(ctx.rows // ctx.ymod, ctx.ymod, -1).sum(dim=0)
        return (d if ctx.needs_input_grad[0] else None), dy


def add_rows(x: Tensor, y: Tensor) -> Tensor:
    if x.shape[0] % y.shape[0]:
        raise ValueError("add_rows: the row count of y must divide that of x")
    return _AddRows.apply(x, y)


class _MeanRows(Function):
    """[groups*rep, D] -> [groups, D]: mean over the rep rows of every group (average pooling over hw)."""

    @staticmethod
    def forward(ctx, x, groups, rep):
        x = _c(x)
        d = x.shape[1]
        out = x.new_empty(groups, d)
        L.colmean(x, groups, rep, d, out, x.new_empty(groups, L.COLMEAN_SPLIT, d))
        ctx.dims = (groups, rep, d)
        return out

    @staticmethod
    def backward(ctx, dy):
        groups, rep, d = ctx.dims
        dx = dy.new_empty(groups * rep, d)
        L.row_broadcast(_c(dy), groups, rep, d, 1.0 / rep, dx)
        return dx, None, None


def mean_rows(x: Tensor, groups: int, rep: int) -> Tensor:
    return _MeanRows.apply(x, groups, rep)


def _conv3x3_raw(x: Tensor, wk: Tensor, b: Optional[Tensor], bsz: int, h: int, wd: int, cin: int, cout: int, split: bool = False) -> Tensor:
    """x [B*H*W, cin], wk [cout, (ky, kx, cin)]: implicit GEMM when cin % 32 == 0, im2col + la_gemm otherwise.  split: the operand is an
    ACTIVATION (O(1) magnitudes) - 32 -> 32 channels may run as three fp16 products on plane pairs (la_conv3x3_split, 4e-7 of fp64 at 3.5x
    the rate of the exact-fp32 MFMA).  Gradients never take that path: fp16 planes bottom out at 6e-8, and the mean-reduced focal
    objective's d loss / d feature entries are 1e-8 ... 1e-5 (measured: 7e-3 ... 2e-2 on the upstream parameter gradients when they did)."""
    y = x.new_empty(x.shape[0], cout)
    if split and L.conv3x3_split_ok(cin, cout) and x.is_contiguous() and wk.is_contiguous():
        L.conv3x3_split(x, bsz, h, wd, cin, wk, b, cout, y)
    elif cin % 32 == 0:
        L.conv3x3_f32(x, bsz, h, wd, cin, wk, b, cout, y)
    else:
        col = x.new_empty(x.shape[0], 9 * cin)
        L.im2col_3x3(x, bsz, h, wd, cin, col)
        L.gemm(col, wk, bias=b, out32=y)
    return y


class _Conv3x3(Function):
    """3x3 / pad 1 convolution on an NHWC map [B*H*W, Cin]; w is the nn.Conv2d weight (Cout, Cin, 3, 3)."""

    @staticmethod
    def forward(ctx, x, w, b, bsz, h, wd):
        x = _c(x)
        cout, cin = w.shape[0], w.shape[1]
        wk = _c(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin))              # [Cout, (ky, kx, cin)]
        y = _conv3x3_raw(x, wk, b, bsz, h, wd, cin, cout, split=True)
        ctx.save_for_backward(x, w)
        ctx.dims = (bsz, h, wd, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bsz, h, wd, has_bias = ctx.dims
        cout, cin = w.shape[0], w.shape[1]
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX = conv3x3(dY, taps flipped, channels transposed)
            wt = _c(w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout))
            dx = _conv3x3_raw(dy, wt, None, bsz, h, wd, cout, cin)
        if ctx.needs_input_grad[1]:
            col = x.new_empty(x.shape[0], 9 * cin)
            L.im2col_3x3(x, bsz, h, wd, cin, col)
            dwk = x.new_zeros(cout, 9 * cin)
            L.gemm_tn(dy, col, dwk)
            dw = dwk.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.new_zeros(cout, 1)
            L.colsum_acc(dy, db)
            db = db.view(-1)
        return dx, dw, db, None, None, None


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], bsz: int, h: int, wd: int) -> Tensor:
    return _Conv3x3.apply(x, w, b, bsz, h, wd)


class _Classify(Function):
    """seg[b, c, pix] = protos[b, c, :] . feat[b, pix, :]."""

    @staticmethod
    def forward(ctx, feat, protos, bsz, npix, c):
        feat, protos = _c(feat), _c(protos)
        cf = feat.shape[1]
        seg = feat.new_empty(bsz, c, npix)
        L.classify(feat, protos, bsz, npix, c, cf, seg)
        ctx.save_for_backward(feat, protos)
        ctx.dims = (bsz, npix, c, cf)
        return seg

    @staticmethod
    def backward(ctx, dseg):
        feat, protos = ctx.saved_tensors
        bsz, npix, c, cf = ctx.dims
        dfeat, dprotos = torch.empty_like(feat), torch.zeros_like(protos)
        L.classify_bwd(_c(dseg), feat, protos, bsz, npix, c, cf, dfeat, dprotos)
        return dfeat, dprotos, None, None, None


def classify(feat: Tensor, protos: Tensor, bsz: int, npix: int, c: int) -> Tensor:
    return _Classify.apply(feat, protos, bsz, npix, c)


class _Bilinear(Function):
    """[N, H, W] -> [N, OH, OW], F.interpolate(mode="bilinear", align_corners=False)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        x = _c(x)
        n, h, w = x.shape
        out = x.new_empty(n, oh, ow)
        L.bilinear(x, n, h, w, oh, ow, out)
        ctx.dims = (n, h, w, oh, ow)
        return out

    @staticmethod
    def backward(ctx, dy):
        n, h, w, oh, ow = ctx.dims
        dy = _c(dy)
        if L.bilinear_bwd_set_ok(oh, ow, h, w):          # a reduction: gathered, dx written (no 1.3 GB zero fill, no atomics on cfg3)
            dx = dy.new_empty(n, h, w)
            L.bilinear_bwd_set(dy, n, oh, ow, oh * ow, ow, dx, h, w, h * w, w)
        else:
            dx = dy.new_zeros(n, h, w)
            L.bilinear_bwd(dy, n, oh, ow, oh * ow, ow, dx, h, w, h * w, w)
        return dx, None, None


def bilinear(x: Tensor, oh: int, ow: int) -> Tensor:
    return _Bilinear.apply(x, oh, ow)


class _RowsToPlanes(Function):
    """[n * hw, c] NHWC rows -> [n * c, hw] planes (and back in the backward): the layout change around a bilinear resize of a dense
    prompt embedding, on the transposer kernels instead of permute().contiguous() (1.3 GB per direction on cfg3)."""

    @staticmethod
    def forward(ctx, x: Tensor, n: int, c: int, hw: int) -> Tensor:
        ctx.dims = (n, c, hw)
        out = torch.empty(n * c, hw, device=x.device)
        L.nhwc_to_nchw(_c(x), n, c, hw, out)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        n, c, hw = ctx.dims
        dx = torch.empty(n * hw, c, device=g.device)
        L.nchw_to_nhwc(_c(g), n, c, hw, out32=dx, dt=L.LA_F32)
        return dx, None, None, None


class _PlanesToRows(Function):
    @staticmethod
    def forward(ctx, x: Tensor, n: int, c: int, hw: int) -> Tensor:
        ctx.dims = (n, c, hw)
        out = torch.empty(n * hw, c, device=x.device)
        L.nchw_to_nhwc(_c(x), n, c, hw, out32=out, dt=L.LA_F32)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        n, c, hw = ctx.dims
        dx = torch.empty(n * c, hw, device=g.device)
        L.nhwc_to_nchw(_c(g), n, c, hw, dx)
        return dx, None, None, None


class _BilinearRows(Function):
    """Bilinear resize of NHWC rows [n * h * w, c] -> [n * oh * ow, c] (la_bilinear_rows) - the dense mask embedding between the mask
    downscaler's 4x-reduced map and the embedding grid (prompt_encoder.py:528-540) without the plane transposes around the planar kernel.
    Backward: reductions are gathered on rows too (la_bilinear_rows_bwd_set); other shapes go through the planar adjoint."""

    @staticmethod
    def forward(ctx, x, n, h, w, c, oh, ow):
        x = _c(x)
        out = x.new_empty(n * oh * ow, c)
        L.bilinear_rows(x, n, h, w, c, oh, ow, out)
        ctx.dims = (n, h, w, c, oh, ow)
        return out

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, oh, ow = ctx.dims
        dy = _c(dy)
        dx = dy.new_empty(n * h * w, c)
        if L.bilinear_bwd_set_ok(oh, ow, h, w):
            L.bilinear_rows_bwd_set(dy, n, oh, ow, c, dx, h, w)
        else:
            planes = dy.new_empty(n * c, oh * ow)
            L.nhwc_to_nchw(dy, n, c, oh * ow, planes)
            dpl = dy.new_zeros(n * c, h, w)
            L.bilinear_bwd(planes, n * c, oh, ow, oh * ow, ow, dpl, h, w, h * w, w)
            L.nchw_to_nhwc(dpl.view(n * c, h * w), n, c, h * w, out32=dx, dt=L.LA_F32)
        return dx, None, None, None, None, None, None


def bilinear_rows(x: Tensor, n: int, h: int, w: int, c: int, oh: int, ow: int) -> Tensor:
    return _BilinearRows.apply(x, n, h, w, c, oh, ow)


def rows_to_planes(x: Tensor, n: int, c: int, hw: int) -> Tensor:
    return _RowsToPlanes.apply(x, n, c, hw)


def planes_to_rows(x: Tensor, n: int, c: int, hw: int) -> Tensor:
    return _PlanesToRows.apply(x, n, c, hw)
