"""Training THROUGH the image encoder (SURVEY 8f row 1 as BASELINE cfg3 names it): forward with saved activations and a hand-written
backward on HIP kernels of the HF ViT stack (``HfEncoderGraph``: transformers ViTModel under models/build_encoder.py:83-100) and of the
SAM ViTDet stack (``SamEncoderGraph``: models/image_encoder.py:110-376).

Which configuration this is: ``parameters/trainval/coco20i/mae_noembs.yaml`` has no ``freeze_backbone``, so
``Lam.get_learnable_params`` (models/lam.py:321-347) returns ``self.parameters()`` - the ViT-B backbone (85.8 M parameters) trains
together with neck, prompt encoder and mask decoder.  (``mae.yaml`` is the decoder-only training on precomputed embeddings.)

Numerics: the forward is the inference engine's (16-bit MFMA operands, split-precision weight planes as ``Lam.precise`` says), run
layer by layer with the activations the backward needs kept in HBM (36 E bytes per token and layer: 7.8 GB per 26-image episode of
ViT-B at 480 px - nothing at 288 GB).  The backward runs its GEMM operands in the same 16-bit type under LOSS SCALING: the gradient
that arrives from the decoder graph is multiplied by a power of two that puts its largest entry near 64, every 16-bit operand and
fp32 intermediate of the encoder backward carries that factor, and the parameter gradients are un-scaled when they are folded into
the flat gradient buffer (non-finite results: the step is repeated with a smaller factor).
  data gradients   dX = dY . W     la_gemm on W^T re-packed per step (16-bit single plane)
  weight gradients dW += dY^T . X  la_gemm_tn16: 16-bit MFMA straight from the row-major operands (LDS transpose reads; the token range cut
                                   into chunks that add into dW with fp32 atomics, db = colsum(dY) on the way) - rounds 3 - 4: la_gemm
                                   with LaGemmEpilogue.ksplit on la_transpose16 copies (``tn16 = False``); shapes neither takes (widths
                                   that are not multiples of 256: the reduced test encoders) go to la_gemm_tn (exact-fp32 MFMA)
  MLP                              the GELU and gelu' run in GEMM epilogues where the shape is on the four-wave kernel
                                   (LaGemmEpilogue.aux16: fc1 writes activation + pre-activation, fc2's dX is multiplied by
                                   gelu'(pre)); else la_gelu_fwd16 / la_gelu_bwd16 passes
  attention        la_attn_fwd_lse / la_attn_bwd (flash form, recomputed probabilities, csrc/attn_bwd.hip)
  LayerNorm        la_layernorm_bwd / la_layernorm_bwd_res
  SAM stack         la_relpos_terms + la_attn_fwd_relpos_lse / la_attn_bwd_relpos + la_relpos_bwd (window and global attention with the
                    decomposed rel-pos bias; gradients of rel_pos_h / rel_pos_w), windows as row gathers - see SamEncoderGraph
Heads: 64 wide (ViT-MAE-B / -L, DINO, IN21k - cfg3, cfg5 -, SAM ViT-B / -L) natively; other widths up to 128 run ZERO-PADDED to 64 / 128
columns per head exactly as in the inference engine (``LamEngine.head_pad``: SAM ViT-H's 80 -> 128, build_encoder.py:9-28; 32-wide test
encoders -> 64): the packed q | k | v weights emit padded heads, the attention kernels (forward and backward, NH = 1 or 2 halves of 64) see
E_attn = heads * padded width, the padded columns of q, k, v, O and of every gradient are exactly zero, and the weight / bias / rel-pos
table gradients are computed on the padded shapes and folded back onto the parameters' own rows and columns.  Rel-pos tables of another
grid: get_rel_pos's resampling is applied as its fixed linear map and that map's transpose.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import _lib as L

Tensor = torch.Tensor


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class HfEncoderGraph:
    """One trainable HF ViT encoder.  ``grads``: parameter name (``image_encoder....``) -> fp32 tensor the gradient is ADDED to."""

    TARGET_MAX = 64.0
    KIND = "hf"

    @staticmethod
    def owns(key: str) -> bool:
        """Parameters whose gradient this graph writes (every ``image_encoder.*`` tensor here)."""
        return key.startswith("image_encoder.")

    def __init__(self, lam, grads: Dict[str, Tensor], precise=None):
        spec = lam.cfg.encoder_spec
        if spec is None or spec.kind != self.KIND:
            raise NotImplementedError(f"{type(self).__name__} is the backward of the {self.KIND!r} encoder stack (HfEncoderGraph: ViT-MAE / DINO / "
                                      "IN21k; SamEncoderGraph: the SAM ViTDet stack)")
        self.hdp = 64 * ((spec.head_dim + 63) // 64)          # head width the attention kernels run (zero-padded columns beyond head_dim)
        if self.hdp > 128:
            raise NotImplementedError(f"encoder backward: heads wider than 128 are not built (got {spec.head_dim})")
        self.lam, self.spec, self.grads = lam, spec, grads
        # private engine: the encoder's packed weights under the TRAINING numerics (precise), re-packed after every optimizer step;
        # the caller's ``lam.precise`` / inference engine are not touched
        self.precise = tuple(lam.precise) if precise is None else tuple(precise)
        self._eng = None
        self._eng_stale = True
        self._seen_version = lam.weights_version      # staleness is driven by the MODEL (load_state_dict / _apply / invalidate bump it)
        self.before_backward = None        # optional callback at the head of ``backward`` (LamTrainer: launch the decoder-side gradient bucket)
        self.w: Dict[str, Tensor] = {k: v for k, v in lam.state_dict(keep_vars=True).items() if k.startswith("image_encoder.")}
        grads = {k: v for k, v in grads.items() if self.owns(k)}
        self.grads = grads
        missing = [k for k, v in self.w.items() if v.is_floating_point() and self.owns(k) and k not in grads]
        if missing:
            raise KeyError(f"no gradient slot for {missing[:3]} ...")
        n = sum(self.w[k].numel() for k in grads)
        dev = lam._device()
        self.scratch = torch.zeros(n, device=dev)               # loss-scaled gradients of one backward pass
        self.sviews: Dict[str, Tensor] = {}
        off = 0
        for k in grads:
            m = self.w[k].numel()
            self.sviews[k] = self.scratch[off:off + m].view_as(self.w[k])
            off += m
        self.ctx = None
        self.last_scale = 1.0
        self.fast_wgrad = True            # 16-bit split-K weight gradients where the shape allows (False: exact-fp32 la_gemm_tn everywhere)
        self.fused_gelu = True            # fc1 forward / fc2 data gradient with the GELU (and gelu') in the GEMM epilogue where the shape allows
        self.tn16 = True                  # ... straight from the row-major 16-bit operands (la_gemm_tn16); False: transposed copies + la_gemm ksplit
        self._tbufs: Dict[tuple, Tensor] = {}
        self._xt_key = None                   # which activation the transposed-operand scratch of _wgrad currently holds
        self._runs = None
        self._b_ready: Dict[tuple, bool] = {}
        self._wt_cache: Dict[str, Tensor] = {}      # 16-bit W^T copies of the data-gradient GEMMs, per optimizer step

    def weights_changed(self) -> None:
        """The optimizer wrote the parameters (through raw pointers): re-pack before the next forward."""
        self._eng_stale = True
        self._wt_cache.clear()

    def _sync_version(self) -> None:
        """A checkpoint restore, best-weights reload, EMA swap or manual edit + ``lam.invalidate()`` reaches this graph through the
        model's ``weights_version``: the private engine and the W^T copies are dropped exactly like after an optimizer step."""
        v = self.lam.weights_version
        if v != self._seen_version:
            self._seen_version = v
            self.weights_changed()

    def engine(self):
        from .engine import LamEngine
        lam = self.lam
        self._sync_version()
        if self._eng is None:
            self._eng = LamEngine(lam.cfg, lam.state_dict(), lam._device(), lam.compute_dtype, lam.decoder_dtype, self.precise, scope="encoder")
        elif self._eng_stale:
            self._eng.repack_encoder(lam.state_dict())
        self._eng_stale = False
        return self._eng

    # ---- forward ---------------------------------------------------------------------------------------------------------------
    def _qkv_plain(self, eng, x16: Tensor, key: str, qkv: Tensor, ea: int) -> None:
        """q | k | v rows of x W^T + b, V row-major as well (the inference path writes V only transposed)."""
        b = eng.p[key[:-2] + ".b"]
        if (key + ".qk") in eng.p:
            L.gemm(x16, eng.p[key + ".qk"], bias=b[: 2 * ea], out16=qkv[:, : 2 * ea])
            L.gemm(x16, eng.p[key + ".v"], bias=b[2 * ea:], out16=qkv[:, 2 * ea:], a_kmod=eng.kmod.get(key + ".v", 0))
        else:
            eng.gemm_w(x16, key, bias=b, out16=qkv)

    @torch.no_grad()
    def forward(self, images: Tensor) -> Tensor:
        """(Bn, 3, S, S) fp32 on the device -> [Bn * hw, E] fp32 NHWC rows (CLS dropped), activations kept for ``backward``."""
        eng = self.engine()
        spec, w, p = self.spec, eng.w32, eng.p
        pre = "image_encoder"
        images = images.contiguous()
        bn, _, s, _ = images.shape
        e, heads, g = spec.dim, spec.heads, s // spec.patch
        hw, t = g * g, g * g + 1
        rows, tpad = bn * t, _ceil(t, 64)
        dev, dt = images.device, eng.dt
        pos = eng._hf_pos(g)
        cls_row = eng._hfpos_cache[-g]
        a, akw = eng.patches("hf.patchA", images, bn * hw, spec.patch)
        res = torch.empty(rows, e, device=dev)
        res.view(bn, t, e)[:, 0].copy_(cls_row)
        L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".embeddings.patch_embeddings.projection.bias"], res=pos, res_mod=t, out32=res,
               map=L.MAP_GROUP, p=(hw, t, 1, 0, 0), **akw)
        scale = spec.head_dim ** -0.5
        layers: List[dict] = []
        hdp = self.hdp
        ea, nh = heads * hdp, hdp // 64             # width of the q / k / v / attention-output blocks (== e unless the heads are padded)
        vt = torch.zeros(bn * heads, hdp, tpad, device=dev, dtype=dt)
        for i in range(spec.depth):
            lp = f"{pre}.encoder.layer.{i}"
            sv = {"x_in": res}                       # (the stream is never updated in place: every residual GEMM writes a new buffer,
            x16 = torch.empty(rows, e, device=dev, dtype=dt)      # which is also the activation the backward keeps)
            sv["xn"] = x16                           # the weight gradients take the 16-bit LayerNorm output the GEMMs saw
            eng.ln(res, lp + ".layernorm_before", 1e-12, out16=x16)
            sv["qkv"] = torch.empty(rows, 3 * ea, device=dev, dtype=dt)
            self._qkv_plain(eng, x16, lp + ".qkv.w", sv["qkv"], ea)
            L.head_transpose(sv["qkv"], 2 * ea, bn, heads * nh, t, tpad, vt)        # (a 128-wide head = two 64-row blocks of V^T)
            sv["ao"] = torch.empty(rows, ea, device=dev, dtype=dt)
            sv["lse"] = torch.full((bn * heads, tpad), 1e30, device=dev)
            L.attn_fwd_lse(sv["qkv"], vt, sv["ao"], sv["lse"], bn, heads, t, tpad, ea, scale)
            x_mid = torch.empty(rows, e, device=dev)
            eng.gemm_w(sv["ao"], lp + ".o.w", bias=w[lp + ".attention.output.dense.bias"], res=res, out32=x_mid)
            sv["x_mid"] = x_mid
            x16b = torch.empty(rows, e, device=dev, dtype=dt)
            sv["xnb"] = x16b
            eng.ln(x_mid, lp + ".layernorm_after", 1e-12, out16=x16b)
            sv["post"] = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
            sv["pre"] = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
            self._fc1_fwd(eng, x16b, lp + ".fc1.w", w[lp + ".intermediate.dense.bias"], sv["pre"], sv["post"])
            # fused / unfused is decided ONCE per graph, here, for the backward half too (la_gemm_fused_act_ok reads the library's
            # mutable kernel selection: an A/B tool that toggles it between forward and backward must not mix the two forms)
            sv["fuse_fc2_bwd"] = bool(self.fused_gelu and L.gemm_fused_act_ok(rows, spec.mlp, e))
            res = torch.empty(rows, e, device=dev)
            eng.gemm_w(sv["post"], lp + ".fc2.w", bias=w[lp + ".output.dense.bias"], res=x_mid, out32=res)
            layers.append(sv)
        fin = torch.empty(rows, e, device=dev)
        eng.ln(res, pre + ".layernorm", 1e-12, out32=fin)
        out = fin.view(bn, t, e)[:, 1:].reshape(bn * hw, e).contiguous()
        self.ctx = dict(images=images, layers=layers, x_fin=res, bn=bn, g=g, hw=hw, t=t, rows=rows, tpad=tpad, e=e, ea=ea, heads=heads,
                        scale=scale, dt=dt)
        return out

    def _fc1_fwd(self, eng, x16: Tensor, key: str, bias: Tensor, pre: Tensor, post: Tensor) -> None:
        """post = GELU(x W1^T + b), pre = x W1^T + b (what gelu' needs in the backward): one launch where the shape runs on the
        persistent four-wave kernel (LaGemmEpilogue.aux16: the epilogue writes both), else the GEMM + la_gelu_fwd16."""
        if self.fused_gelu and eng.kmod.get(key, 0) == 0 and L.gemm_fused_act_ok(x16.shape[0], pre.shape[1], x16.shape[1]):
            L.gemm(x16, eng.p[key], bias=bias, out16=post, act=L.ACT_GELU, aux16=pre)
        else:
            eng.gemm_w(x16, key, bias=bias, out16=pre)
            L.gelu_fwd16(pre, post)

    def _fc2_bwd(self, dy32: Tensor, dy16: Tensor, a: dict, wname: str, bname: str, dh: Tensor, dpre32, dpre16: Tensor) -> None:
        """Backward of fc2 and of the GELU in front of it: dW2 += dY^T post, db2 += colsum dY, d pre = (dY W2) * gelu'(pre) - the last as
        ONE product whose epilogue reads the saved pre-activation (LA_ACT_GELU_BWD) where the shape allows, else product + la_gelu_bwd16.
        dpre32 (None = not needed): the fp32 copy the exact-fp32 fallbacks of fc1's weight / bias gradient read."""
        r, n, k = dy16.shape[0], a["pre"].shape[1], dy16.shape[1]
        fuse = a["fuse_fc2_bwd"] if "fuse_fc2_bwd" in a else (self.fused_gelu and L.gemm_fused_act_ok(r, n, k))
        if fuse and dpre32 is None:
            wt = self.w[wname]
            if not self._wgrad(dy16, dy32, a["post"], self.sviews[wname], db=self.sviews[bname]):
                L.colsum_acc(dy32, self.sviews[bname])
            L.gemm(dy16, self._wt16(wname, lambda: wt, dy16.dtype), out16=dpre16, act=L.ACT_GELU_BWD, aux16=a["pre"])
        else:
            self._linear_bwd(dy32, dy16, a["post"], wname, bname, dx32=dh)
            L.gelu_bwd16(a["pre"], dh, dpre32, dpre16)

    # ---- backward --------------------------------------------------------------------------------------------------------------
    def _wt16(self, key: str, wt_fn, dt) -> Tensor:
        """nn.Linear weight [N, K] fp32 -> W^T [K, N] 16-bit: the 'weight' of the data-gradient GEMM dX = dY . W.  One copy per
        optimizer step (``weights_changed`` drops them): a loss-scale retry or a gradient-accumulation micro-step re-uses it."""
        self._sync_version()
        out = self._wt_cache.get(key)
        if out is None or out.dtype != dt:
            wt = wt_fn()
            n, k = wt.shape
            out = torch.empty(k, n, device=wt.device, dtype=dt)
            L.nchw_to_nhwc(wt.detach().contiguous(), 1, n, k, out16=out, dt=L._DT[dt])
            self._wt_cache[key] = out
        return out

    def _wgrad(self, dy16, dy32, x, dw: Tensor, db: Optional[Tensor] = None) -> bool:
        """dw[N, K] += dy[R, N]^T x[R, K].  dy16 / dy32: the same gradient in 16 bit and fp32 (either may be None; column slices
        allowed); x: fp32 or 16-bit.  db: the layer's bias gradient [N]; returns True when db += colsum(dy) was done on the way (the
        transpose of dy for the split-K product reads every element anyway)."""
        dy = dy16 if dy16 is not None else dy32
        r, n = dy.shape
        k = x.shape[1]
        if self.fast_wgrad and self.tn16 and dy16 is not None and x.dtype == dy16.dtype and n % 256 == 0 and k % 256 == 0 and r >= 128:
            # both operands already 16-bit and row-major: the product reads them as they are (LDS transpose reads), db on the way
            fused = db is not None and db.is_contiguous()
            L.gemm_tn16(dy16, x, dw.view(n, k), db=db if fused else None)
            return fused
        if self.fast_wgrad and k % 256 == 0 and r >= 128:
            rp = _ceil(r, 64)
            dt = self.ctx["dt"]
            dyt = self._tbuf("dyt", n, rp, dt)
            xt = self._tbuf("xt", k, rp, dt)
            fused = db is not None and db.is_contiguous()
            L.transpose16(dy, dyt, colsum=db if fused else None)
            xkey = (x.data_ptr(), tuple(x.shape), x.dtype)
            if self._xt_key != xkey:                 # q, k and v share their input: one transposed copy serves the three products
                L.transpose16(x, xt)
                self._xt_key = xkey
            L.gemm(dyt, xt, out32=dw.view(n, k), ksplit=1)
            return fused
        else:
            x32 = x
            if x.dtype != torch.float32:
                x32 = self._tbuf("x32", r, k, torch.float32)
                L.cast(x.contiguous(), x32)
            if dy32 is None:
                dy32 = self._tbuf("dy32", r, n, torch.float32)
                L.cast(dy16.contiguous(), dy32)
            L.gemm_tn(dy32, x32, dw.view(n, k))
            return False

    def _qkv_wgrad_fused(self, dqkv16: Tensor, x: Tensor, att: str, e: int) -> bool:
        """The three weight gradients of HF's separate query / key / value Linears from ONE split-K product dqkv^T x.  In the flat
        gradient buffer (FlatAdamW packs the tensors back to back in parameter order) the three [E, E] blocks sit a fixed stride
        apart - weight, bias, weight, bias, ... - which is la_gemm's LA_MAP_GROUP row map; the bias gradients come out of the
        transpose's column sums.  Three launches of 9 output tiles each end in three times the atomics of one launch of 27.
        Returns False (nothing done) when the layout or the shape does not allow it."""
        sv = self.sviews
        gw = [sv[att + nm + ".weight"] for nm in ("query", "key", "value")]
        gb = [sv[att + nm + ".bias"] for nm in ("query", "key", "value")]
        r = dqkv16.shape[0]
        if not (self.fast_wgrad and e % 256 == 0 and r >= 128 and all(g.is_contiguous() for g in gw + gb)):
            return False
        stride = gw[1].data_ptr() - gw[0].data_ptr()
        if stride <= 0 or stride % (4 * e) or gw[2].data_ptr() - gw[1].data_ptr() != stride:
            return False
        rows_apart = stride // (4 * e)                       # in rows of E floats: E weight rows + the bias row(s) in between
        if self.tn16 and x.dtype == dqkv16.dtype:
            cs = self._tbuf("qkv_colsum", 3, e, torch.float32)
            cs.zero_()
            L.gemm_tn16(dqkv16, x, gw[0].view(e, e), db=cs.view(-1), gsize=e, gstride=rows_apart)
            for j in range(3):
                gb[j].add_(cs[j])
            return True
        rp = _ceil(r, 64)
        dt = self.ctx["dt"]
        dyt = self._tbuf("dyt", 3 * e, rp, dt)
        xt = self._tbuf("xt", e, rp, dt)
        cs = self._tbuf("qkv_colsum", 3, e, torch.float32)
        cs.zero_()
        L.transpose16(dqkv16, dyt, colsum=cs.view(-1))
        xkey = (x.data_ptr(), tuple(x.shape), x.dtype)
        if self._xt_key != xkey:
            L.transpose16(x, xt)
            self._xt_key = xkey
        L.gemm(dyt, xt, out32=gw[0].view(e, e), ksplit=1, map=L.MAP_GROUP, p=(e, rows_apart, 0, 0, 0))
        for j in range(3):
            gb[j].add_(cs[j])
        return True

    def _tbuf(self, name: str, a: int, b: int, dtype) -> Tensor:
        key = (name, a, b, dtype)
        t = self._tbufs.get(key)
        if t is None:
            t = torch.empty(a, b, device=self.scratch.device, dtype=dtype)
            self._tbufs[key] = t
        return t

    def _linear_bwd(self, dy32: Tensor, dy16: Tensor, x, wname: str, bname: str, dx32=None, dx16=None, pad_in: bool = False,
                    pad_out: int = 0) -> None:
        """dW += dY^T X, db += colsum(dY), dX = dY W for one nn.Linear (weight ``wname`` [N, K]); x: the layer's input, any dtype.
        pad_in: x carries zero-padded heads (the attention output); pad_out = number of head blocks of dY that are zero-padded (q | k | v)."""
        wt = self.w[wname]
        if pad_in or pad_out:
            wt_fn = lambda: self._pad_weight(wt, pad_in, pad_out)           # noqa: E731
            n, k = dy16.shape[1], x.shape[1]
            dwp = self._tbuf("dw_pad", n, k, torch.float32)
            dbp = self._tbuf("db_pad", 1, n, torch.float32).view(n)
            dwp.zero_()
            dbp.zero_()
        else:
            wt_fn = lambda: wt                                              # noqa: E731
            dwp, dbp = self.sviews[wname], self.sviews[bname]
        if not self._wgrad(dy16, dy32, x, dwp, db=dbp):
            if dy32 is None:
                dy32 = self._tbuf("dy32b", dy16.shape[0], dy16.shape[1], torch.float32)
                L.cast(dy16.contiguous(), dy32)
            L.colsum_acc(dy32, dbp)
        if pad_in or pad_out:
            self._fold_padded(dwp, dbp, self.sviews[wname], self.sviews[bname], pad_in, pad_out)
        L.gemm(dy16, self._wt16(wname, wt_fn, dy16.dtype), out32=dx32, out16=dx16)

    # ---- zero-padded heads (head_dim not a multiple of 64) -----------------------------------------------------------------------
    def _pad_weight(self, wt: Tensor, pad_in: bool, pad_out: int) -> Tensor:
        """The nn.Linear weight as the padded-head GEMMs see it: zero rows appended to each of the ``pad_out`` head blocks of the output
        dimension (q | k | v producers), zero columns to each head block of the input dimension (``pad_in``: the attention-output consumer)."""
        from .engine import LamEngine
        hd, hdp, heads = self.spec.head_dim, self.hdp, self.spec.heads
        wt = wt.detach().reshape(wt.shape[0], -1)
        if pad_out:
            wt = LamEngine._pad_heads_out(wt, pad_out, hd, hdp)
        if pad_in:
            wt = LamEngine._pad_heads_in(wt, heads, hd, hdp)
        return wt

    def _fold_padded(self, dwp: Tensor, dbp: Optional[Tensor], dw: Tensor, db: Optional[Tensor], pad_in: bool, pad_out: int) -> None:
        """Gradients computed on the padded shapes -> the parameters' own rows / columns (the padded ones multiply zeros: dropped)."""
        hd, hdp, heads = self.spec.head_dim, self.hdp, self.spec.heads
        g = dwp
        if pad_out:
            g = g.view(pad_out, hdp, -1)[:, :hd].reshape(pad_out * hd, -1)
            if dbp is not None:
                db.view(-1).add_(dbp.view(pad_out, hdp)[:, :hd].reshape(-1))
        elif dbp is not None:
            db.view(-1).add_(dbp.view(-1))
        if pad_in:
            g = g.view(g.shape[0], heads, hdp)[:, :, :hd].reshape(g.shape[0], heads * hd)
        dw.view(g.shape).add_(g)

    @torch.no_grad()
    def backward(self, d_out: Tensor) -> None:
        """d_out: gradient w.r.t. ``forward``'s result [Bn * hw, E] fp32.  Adds every encoder parameter's gradient to ``grads``."""
        if self.before_backward is not None:
            self.before_backward()
        lo, hi = (float(v) for v in torch.stack(torch.aminmax(d_out)).tolist())      # one pass, one read-back (min / max propagate NaN)
        amax = max(abs(lo), abs(hi)) if lo == lo and hi == hi else float("nan")
        if not math.isfinite(amax):
            raise FloatingPointError("non-finite gradient at the encoder output")
        if amax == 0.0:
            return
        scale = 2.0 ** math.floor(math.log2(self.TARGET_MAX / amax))
        for _ in range(4):
            self.scratch.zero_()
            self._backward_scaled(d_out, scale)
            # every scaled gradient finite?  (a sum can overflow or cancel to a finite value; isfinite().all() is five elementwise passes
            # over the 86 M entries - 0.7 ms - where ONE min / max reduction, which propagates NaN and shows +-inf, says the same)
            if bool(torch.isfinite(torch.stack(torch.aminmax(self.scratch))).all()):
                break
            scale /= 256.0                 # a 16-bit intermediate overflowed: repeat with more head-room
        else:
            raise FloatingPointError("encoder backward overflowed at every loss scale tried")
        self.last_scale = scale
        for off, m, dst in self._unscale_runs():
            L.axpy(self.scratch[off:off + m], dst, 1.0 / scale)
        self.ctx = None

    def _unscale_runs(self):
        """(scratch offset, length, destination view) per run of gradient slots that are adjacent in memory, in scratch order - the
        optimizer's flat buffer keeps the encoder tensors together, so the un-scaling is one launch, not one per tensor."""
        if self._runs is None:
            runs, off = [], 0
            for gslot in self.grads.values():
                m, flat = gslot.numel(), gslot.view(-1)
                if runs and flat.data_ptr() == runs[-1][2].data_ptr() + 4 * runs[-1][1] and flat.untyped_storage().data_ptr() == \
                        runs[-1][2].untyped_storage().data_ptr():
                    runs[-1][1] += m
                else:
                    runs.append([off, m, flat])
                off += m
            self._runs = [(o, m, torch.as_strided(f, (m,), (1,))) for o, m, f in runs]
        return self._runs

    def _backward_scaled(self, d_out: Tensor, s: float) -> None:
        c = self.ctx
        if c is None:
            raise RuntimeError("HfEncoderGraph.backward without a forward")
        self._xt_key = None
        spec, w, sv = self.spec, self.w, self.sviews
        pre = "image_encoder"
        bn, t, hw, rows, tpad, e, heads, dt = c["bn"], c["t"], c["hw"], c["rows"], c["tpad"], c["e"], c["heads"], c["dt"]
        ea, padded = c["ea"], c["ea"] != c["e"]
        dev = d_out.device
        dfin = torch.zeros(rows, e, device=dev)
        tmp = torch.empty_like(d_out)
        L.cast(d_out.contiguous(), tmp, s)
        dfin.view(bn, t, e)[:, 1:].copy_(tmp.view(bn, hw, e))
        # dres: the running fp32 gradient of the residual stream, d16 its 16-bit copy (the operand of the next backward GEMM) - both
        # leave every LayerNorm backward together with the skip connection's share (la_layernorm_bwd_res)
        dres = torch.empty(rows, e, device=dev)
        d16 = torch.empty(rows, e, device=dev, dtype=dt)
        L.layernorm_bwd_res(c["x_fin"], dfin, w[pre + ".layernorm.weight"], w[pre + ".layernorm.bias"], 1e-12, None, dres, d16,
                            sv[pre + ".layernorm.weight"], sv[pre + ".layernorm.bias"])
        dh = torch.empty(rows, spec.mlp, device=dev)
        dpre32 = torch.empty(rows, spec.mlp, device=dev)
        dpre16 = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
        dxn = torch.empty(rows, e, device=dev)
        dao = torch.empty(rows, ea, device=dev, dtype=dt)
        dqkv16 = torch.empty(rows, 3 * ea, device=dev, dtype=dt)
        dqkv32 = torch.empty(rows, 3 * ea, device=dev)
        dvec = torch.zeros(bn * heads, tpad, device=dev)
        for i in reversed(range(spec.depth)):
            lp = f"{pre}.encoder.layer.{i}"
            a = c["layers"][i]
            # ---- MLP: res = x_mid + fc2(gelu(fc1(LN2(x_mid)))) -----------------------------------------------------------------
            # (the fp32 copy of d pre-activation only feeds the exact-fp32 fallbacks of the weight / bias gradient)
            need32 = not (self.fast_wgrad and e % 256 == 0 and rows >= 128)
            self._fc2_bwd(dres, d16, a, lp + ".output.dense.weight", lp + ".output.dense.bias", dh, dpre32 if need32 else None, dpre16)
            self._linear_bwd(dpre32 if need32 else None, dpre16, a["xnb"], lp + ".intermediate.dense.weight", lp + ".intermediate.dense.bias",
                             dx32=dxn)
            L.layernorm_bwd_res(a["x_mid"], dxn, w[lp + ".layernorm_after.weight"], w[lp + ".layernorm_after.bias"], 1e-12, dres, dres, d16,
                                sv[lp + ".layernorm_after.weight"], sv[lp + ".layernorm_after.bias"])
            # ---- attention: x_mid = x_in + proj(attn(LN1(x_in))) -------------------------------------------------------------
            self._linear_bwd(dres, d16, a["ao"], lp + ".attention.output.dense.weight", lp + ".attention.output.dense.bias", dx16=dao,
                             pad_in=padded)
            # (no K^T / Q^T / dO^T copies: the backward kernels read those operands out of the row-major tiles with LDS transpose reads)
            L.attn_bwd(a["qkv"], a["ao"], dao, None, None, None, a["lse"], dvec, dqkv16, bn, heads, t, tpad, ea, c["scale"])
            att = lp + ".attention.attention."
            have32 = False
            fused = not padded and self._qkv_wgrad_fused(dqkv16, a["xn"], att, e)
            for j, nm in enumerate(() if fused else ("query", "key", "value")):
                if padded:
                    dwp = self._tbuf("dw_pad_qkv", ea, e, torch.float32)
                    dbp = self._tbuf("db_pad_qkv", 1, ea, torch.float32).view(ea)
                    dwp.zero_()
                    dbp.zero_()
                else:
                    dwp, dbp = sv[att + nm + ".weight"], sv[att + nm + ".bias"]
                if not self._wgrad(dqkv16[:, j * ea:(j + 1) * ea], None, a["xn"], dwp, db=dbp):
                    if not have32:                    # (exact-fp32 fallback of the weight gradient: it needs the fp32 copy)
                        L.cast(dqkv16, dqkv32)
                        have32 = True
                    L.colsum_acc(dqkv32[:, j * ea:(j + 1) * ea], dbp)
                if padded:
                    self._fold_padded(dwp, dbp, sv[att + nm + ".weight"], sv[att + nm + ".bias"], False, heads)
            wqkv_t = self._wt16(att + "qkv", lambda: torch.cat([self._pad_weight(w[att + nm + ".weight"], False, heads if padded else 0)
                                                                for nm in ("query", "key", "value")]), dt)
            L.gemm(dqkv16, wqkv_t, out32=dxn)                                                                  # [3E, E]^T
            L.layernorm_bwd_res(a["x_in"], dxn, w[lp + ".layernorm_before.weight"], w[lp + ".layernorm_before.bias"], 1e-12, dres, dres, d16,
                                sv[lp + ".layernorm_before.weight"], sv[lp + ".layernorm_before.bias"])
        # ---- embeddings: res0[b, 0] = cls + pos[0]; res0[b, 1 + i] = patch_i W^T + b + pos[1 + i] --------------------------------
        emb = pre + ".embeddings."
        d0 = dres.view(bn, t, e)
        dpatch = d0[:, 1:].reshape(bn * hw, e).contiguous()
        k = 3 * spec.patch * spec.patch
        a32 = torch.empty(bn * hw, k, device=dev)
        L.im2col_patch(c["images"], spec.patch, a32)
        self._wgrad(None, dpatch, a32, sv[emb + "patch_embeddings.projection.weight"])
        L.colsum_acc(dpatch, sv[emb + "patch_embeddings.projection.bias"])
        dpos_rows = d0.sum(dim=0)                                   # [t, E]: a few hundred rows of bookkeeping, not a kernel
        sv[emb + "cls_token"].view(-1).add_(dpos_rows[0])
        pos = w[emb + "position_embeddings"]
        if c["g"] == spec.pos_grid:
            sv[emb + "position_embeddings"].view(t, e).add_(dpos_rows)
        else:                                                       # through the bicubic resample of the patch positions: dpos = B^T d
            from .engine import bicubic_matrix_t
            bt = bicubic_matrix_t(spec.pos_grid, c["g"], dev)       # [gin^2, gout^2]
            b = self._tbuf("bicubic_b", c["g"] * c["g"], spec.pos_grid * spec.pos_grid, torch.float32)
            if not self._b_ready.get((spec.pos_grid, c["g"])):
                b.copy_(bt.t())
                self._b_ready[(spec.pos_grid, c["g"])] = True
            gp = sv[emb + "position_embeddings"].view(-1, e)        # [1 + gin^2, E]
            gp[0].add_(dpos_rows[0])
            L.gemm_tn(b, dpos_rows[1:].contiguous(), gp[1:])


class SamEncoderGraph(HfEncoderGraph):
    """One trainable SAM ViTDet block stack (models/image_encoder.py:110-131 without the neck; blocks :134-197, attention :200-255, window
    partition :258-304, decomposed relative positions :307-376): ``lam_b`` / ``lam_l`` / ``lam_h``-style models with nothing frozen
    (models/lam.py:321-347).  Same machinery as the HF stack (16-bit operands, loss-scaled backward, split-K weight gradients); what is
    different:
      * window blocks: LayerNorm output is zero-padded to a multiple of the window AFTER the norm and cut into windows (the training
        forward keeps the padded rows: their q / k / v are the bias, they are attended to, their outputs are dropped - exactly the
        reference); the partition / un-partition are row gathers with an index built once per geometry;
      * attention carries the decomposed rel-pos bias: la_relpos_terms -> la_attn_fwd_relpos_lse forward, la_attn_bwd_relpos +
        la_relpos_bwd backward (d rel_pos_h / d rel_pos_w accumulate over images, heads and query rows);
      * no CLS token, a learned absolute position embedding of the full grid, LayerNorm eps 1e-6.
    The SAM neck (1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d) is NOT in this graph: the trainer runs it through the decoder graph's
    autograd operators (``DecoderGraph.conv_neck``), so ``image_encoder.neck.*`` is excluded from ``owns``."""

    KIND = "sam"

    @staticmethod
    def owns(key: str) -> bool:
        return key.startswith("image_encoder.") and not key.startswith("image_encoder.neck.")

    def _win_index(self, bn: int, g: int, ws: int, dev) -> Tensor:
        """Row of image-order token (b, y, x) in the zero-padded, window-partitioned layout [bn * nw * nw * ws * ws]."""
        key = ("widx", bn, g, ws)
        t = self._tbufs.get(key)
        if t is None:
            nw = (g + ws - 1) // ws
            b = torch.arange(bn).view(bn, 1, 1)
            y = torch.arange(g).view(1, g, 1)
            x = torch.arange(g).view(1, 1, g)
            t = ((((b * nw + y // ws) * nw + x // ws) * ws + y % ws) * ws + x % ws).reshape(-1).to(dev)
            self._tbufs[key] = t
        return t

    def _rel_tables(self, bp: str, gg: int):
        """The fp32 rel-pos tables [(2 gg - 1), hd] the backward differentiates through, the buffers ``la_relpos_bwd`` accumulates their
        gradients into, and the step that folds those into the parameters' gradient slots.

        Table length == 2 gg - 1: the parameters and their slots themselves.  Otherwise ``get_rel_pos`` (image_encoder.py:321-330)
        resamples the table with ``F.interpolate(mode="linear")`` - a fixed linear map R [(2 gg - 1), L] per (L, gg), built once by
        pushing the identity through the same call (as ``bicubic_matrix_t`` does for the HF position table): the backward runs on
        R . table and adds R^T . d(R . table) to the slot (exact-fp32 MFMA products, la_gemm_tn)."""
        w, sv = self.w, self.sviews
        tabs, dtabs, folds, unpads = [], [], [], []
        hdp = self.hdp

        def padded(tab: Tensor, slot: Tensor):
            """Zero-padded heads: the kernels walk tables [(2 gg - 1), hdp]; the gradient's first head_dim columns are the table's."""
            if tab.shape[1] == hdp:
                return tab, slot
            tp = torch.nn.functional.pad(tab, (0, hdp - tab.shape[1])).contiguous()
            dp = torch.zeros_like(tp)
            unpads.append((dp, slot))
            return tp, dp

        for ax in ("h", "w"):
            key = f"{bp}.attn.rel_pos_{ax}"
            tab = w[key].detach()
            ln, hd = tab.shape
            if ln == 2 * gg - 1:
                tp, dp = padded(tab.contiguous(), sv[key])
                tabs.append(tp)
                dtabs.append(dp)
                continue
            rkey = ("relR", ln, gg)
            r = self._tbufs.get(rkey)
            if r is None:                            # R [(2 gg - 1), L] and R^T [L, (2 gg - 1)], constant per geometry
                import torch.nn.functional as F
                eye = torch.eye(ln, dtype=torch.float32).unsqueeze(0)                                  # (1, L "channels", L)
                rt = F.interpolate(eye, size=2 * gg - 1, mode="linear")[0].contiguous().to(tab.device)   # [L, 2 gg - 1] = R^T
                r = (rt.t().contiguous(), rt)
                self._tbufs[rkey] = r
            used = torch.zeros(2 * gg - 1, hd, device=tab.device)
            L.gemm_tn(r[1], tab.contiguous(), used)                  # used = (R^T)^T tab = R . table
            dused = torch.zeros_like(used)
            tp, dp = padded(used, dused)
            tabs.append(tp)
            dtabs.append(dp)
            folds.append((r[0], dused, sv[key]))

        def fold():
            for dp, slot in unpads:
                slot.add_(dp[:, : slot.shape[1]])
            for rm, du, slot in folds:
                L.gemm_tn(rm, du, slot)                               # slot [L, hd] += R^T . d(R . table)
        return tabs, dtabs, fold

    @torch.no_grad()
    def forward(self, images: Tensor) -> Tensor:
        """(Bn, 3, S, S) fp32 on the device -> the last block's state [Bn * hw, E] fp32 NHWC rows, activations kept for ``backward``."""
        eng = self.engine()
        spec, w, p = self.spec, eng.w32, eng.p
        pre = "image_encoder"
        images = images.contiguous()
        bn, _, s, _ = images.shape
        e, heads, g, ws = spec.dim, spec.heads, s // spec.patch, spec.window
        if g != spec.pos_grid:
            raise ValueError(f"the SAM stack has a fixed position embedding: image side {spec.img_size}, got {s}")
        hw, rows = g * g, bn * g * g
        dev, dt = images.device, eng.dt
        a, akw = eng.patches("enc.patchA", images, rows, spec.patch)
        res = torch.empty(rows, e, device=dev)
        L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".patch_embed.proj.bias"], res=p[pre + ".pos"], res_mod=hw, out32=res, **akw)
        scale = spec.head_dim ** -0.5
        hdp = self.hdp
        ea, nh = heads * hdp, hdp // 64             # (zero-padded heads: see HfEncoderGraph.forward)
        nw = (g + ws - 1) // ws
        layers: List[dict] = []
        for i in range(spec.depth):
            bp = f"{pre}.blocks.{i}"
            is_global = i in spec.global_idx
            nb, gg = (bn, g) if is_global else (bn * nw * nw, ws)
            t = gg * gg
            arows, tpad = nb * t, _ceil(t, 64)
            # (rel-pos tables whose length is not 2 gg - 1 are resampled like get_rel_pos does, image_encoder.py:321-330: the engine packs
            # the resampled 16-bit tables the forward kernels read; the backward applies the transposed map, ``_rel_tables``)
            sv = {"x_in": res, "global": is_global, "nb": nb, "g": gg, "t": t, "tpad": tpad, "arows": arows}
            x16 = torch.empty(rows, e, device=dev, dtype=dt)
            eng.ln(res, bp + ".norm1", 1e-6, out16=x16)
            if is_global:
                xa = x16
            else:                                    # pad-after-norm windows: padded rows are zero
                xa = torch.zeros(arows, e, device=dev, dtype=dt)
                xa.index_copy_(0, self._win_index(bn, g, ws, dev), x16)
            sv["xa"] = xa
            sv["qkv"] = torch.empty(arows, 3 * ea, device=dev, dtype=dt)
            self._qkv_plain(eng, xa, bp + ".qkv.w", sv["qkv"], ea)
            sv["relh"] = torch.empty(nb * heads, t, gg, device=dev)
            sv["relw"] = torch.empty(nb * heads, t, gg, device=dev)
            L.relpos_terms(sv["qkv"], nb, heads, gg, ea, p[bp + ".tabh"], p[bp + ".tabw"], sv["relh"], sv["relw"])
            vt = torch.zeros(nb * heads, hdp, tpad, device=dev, dtype=dt)
            L.head_transpose(sv["qkv"], 2 * ea, nb, heads * nh, t, tpad, vt)
            sv["ao"] = torch.empty(arows, ea, device=dev, dtype=dt)
            sv["lse"] = torch.empty(nb * heads, tpad, device=dev)
            L.attn_fwd_relpos_lse(sv["qkv"], vt, sv["ao"], sv["relh"], sv["relw"], sv["lse"], nb, heads, t, tpad, gg, ea, scale)
            x_mid = torch.empty(rows, e, device=dev)
            if is_global:
                eng.gemm_w(sv["ao"], bp + ".proj.w", bias=w[bp + ".attn.proj.bias"], res=res, out32=x_mid)
            else:                                    # proj on every window row, real tokens gathered back into image order
                yw = torch.empty(arows, e, device=dev)
                eng.gemm_w(sv["ao"], bp + ".proj.w", bias=w[bp + ".attn.proj.bias"], out32=yw)
                L.add_cast(res, yw.index_select(0, self._win_index(bn, g, ws, dev)), rows, out32=x_mid, dt=L._DT[dt])
            sv["x_mid"] = x_mid
            x16b = torch.empty(rows, e, device=dev, dtype=dt)
            sv["xnb"] = x16b
            eng.ln(x_mid, bp + ".norm2", 1e-6, out16=x16b)
            sv["post"] = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
            sv["pre"] = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
            self._fc1_fwd(eng, x16b, bp + ".lin1.w", w[bp + ".mlp.lin1.bias"], sv["pre"], sv["post"])
            sv["fuse_fc2_bwd"] = bool(self.fused_gelu and L.gemm_fused_act_ok(rows, spec.mlp, e))      # (decided with the forward: see HfEncoderGraph)
            res = torch.empty(rows, e, device=dev)
            eng.gemm_w(sv["post"], bp + ".lin2.w", bias=w[bp + ".mlp.lin2.bias"], res=x_mid, out32=res)
            layers.append(sv)
        self.ctx = dict(images=images, layers=layers, bn=bn, g=g, hw=hw, rows=rows, e=e, ea=ea, heads=heads, scale=scale, dt=dt, ws=ws)
        return res

    def _backward_scaled(self, d_out: Tensor, s: float) -> None:
        c = self.ctx
        if c is None:
            raise RuntimeError("SamEncoderGraph.backward without a forward")
        self._xt_key = None
        spec, w, sv = self.spec, self.w, self.sviews
        pre = "image_encoder"
        bn, g, hw, rows, e, heads, dt, ws = c["bn"], c["g"], c["hw"], c["rows"], c["e"], c["heads"], c["dt"], c["ws"]
        ea, padded = c["ea"], c["ea"] != c["e"]
        dev = d_out.device
        dres = torch.empty(rows, e, device=dev)
        L.cast(d_out.contiguous(), dres, s)
        d16 = torch.empty(rows, e, device=dev, dtype=dt)
        L.cast(dres, d16)
        dh = torch.empty(rows, spec.mlp, device=dev)
        dpre32 = torch.empty(rows, spec.mlp, device=dev)
        dpre16 = torch.empty(rows, spec.mlp, device=dev, dtype=dt)
        dxn = torch.empty(rows, e, device=dev)
        for i in reversed(range(spec.depth)):
            bp = f"{pre}.blocks.{i}"
            a = c["layers"][i]
            nb, gg, t, tpad, arows = a["nb"], a["g"], a["t"], a["tpad"], a["arows"]
            # ---- MLP: res = x_mid + lin2(gelu(lin1(LN2(x_mid)))) -----------------------------------------------------------------
            need32 = not (self.fast_wgrad and e % 256 == 0 and rows >= 128)
            self._fc2_bwd(dres, d16, a, bp + ".mlp.lin2.weight", bp + ".mlp.lin2.bias", dh, dpre32 if need32 else None, dpre16)
            self._linear_bwd(dpre32 if need32 else None, dpre16, a["xnb"], bp + ".mlp.lin1.weight", bp + ".mlp.lin1.bias", dx32=dxn)
            L.layernorm_bwd_res(a["x_mid"], dxn, w[bp + ".norm2.weight"], w[bp + ".norm2.bias"], 1e-6, dres, dres, d16, sv[bp + ".norm2.weight"],
                                sv[bp + ".norm2.bias"])
            # ---- attention: x_mid = x_in + unpartition(proj(attn(partition(LN1(x_in))))) ------------------------------------------
            if a["global"]:
                dy16, dy32 = d16, dres
            else:                                    # the dropped (padded) output rows receive no gradient
                widx = self._win_index(bn, g, ws, dev)
                dy16 = torch.zeros(arows, e, device=dev, dtype=dt)
                dy16.index_copy_(0, widx, d16)
                dy32 = None
            dao = torch.empty(arows, ea, device=dev, dtype=dt)
            self._linear_bwd(dy32, dy16, a["ao"], bp + ".attn.proj.weight", bp + ".attn.proj.bias", dx16=dao, pad_in=padded)
            dvec = torch.empty(nb * heads, tpad, device=dev)
            dqkv16 = torch.empty(arows, 3 * ea, device=dev, dtype=dt)
            drelh, drelw = torch.empty_like(a["relh"]), torch.empty_like(a["relw"])
            L.attn_bwd_relpos(a["qkv"], a["ao"], dao, None, None, None, a["lse"], dvec, dqkv16, a["relh"], a["relw"], drelh, drelw, nb, heads, t,
                              tpad, gg, ea, c["scale"])
            tabs, dtabs, fold = self._rel_tables(bp, gg)
            L.relpos_bwd(a["qkv"], dqkv16, drelh, drelw, tabs[0], tabs[1], dtabs[0], dtabs[1], nb, heads, gg, ea)
            fold()
            dxa = torch.empty(arows, e, device=dev)
            self._linear_bwd(None, dqkv16, a["xa"], bp + ".attn.qkv.weight", bp + ".attn.qkv.bias", dx32=dxa, pad_out=3 * heads if padded else 0)
            dxn_i = dxa if a["global"] else dxa.index_select(0, self._win_index(bn, g, ws, dev))
            L.layernorm_bwd_res(a["x_in"], dxn_i.contiguous(), w[bp + ".norm1.weight"], w[bp + ".norm1.bias"], 1e-6, dres, dres, d16,
                                sv[bp + ".norm1.weight"], sv[bp + ".norm1.bias"])
        # ---- patch embedding + absolute position embedding: res0[b, i] = patch_i W^T + b + pos[i] -------------------------------------
        k = 3 * spec.patch * spec.patch
        a32 = torch.empty(rows, k, device=dev)
        L.im2col_patch(c["images"], spec.patch, a32)
        self._wgrad(None, dres, a32, sv[pre + ".patch_embed.proj.weight"].view(e, k))
        L.colsum_acc(dres, sv[pre + ".patch_embed.proj.bias"])
        sv[pre + ".pos_embed"].view(hw, e).add_(dres.view(bn, hw, e).sum(dim=0))        # (a few thousand rows of bookkeeping)


class _EncoderFn(torch.autograd.Function):
    """Autograd shell: forward / backward of ``HfEncoderGraph`` around the decoder graph's autograd."""

    @staticmethod
    def forward(ctx, graph: HfEncoderGraph, images: Tensor, anchor: Tensor):
        ctx.graph = graph
        return graph.forward(images)

    @staticmethod
    def backward(ctx, g):
        ctx.graph.backward(g.contiguous())
        return None, None, None


def encode_trainable(graph: HfEncoderGraph, images: Tensor, anchor: Tensor) -> Tensor:
    return _EncoderFn.apply(graph, images, anchor)
