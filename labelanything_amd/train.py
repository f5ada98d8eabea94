"""Training step of the hot path: forward -> focal objective -> backward -> gradient all-reduce -> AdamW.

Mirror of the reference's ``WrapperModule`` (experiment/utils.py:266-303: ``model(input_dict)`` then ``LabelAnythingLoss``) and of
one iteration of ``Run.train_epoch`` (experiment/run.py:500-568: forward, ``accelerator.backward(loss / normalizer)``,
``optimizer.step()``, scheduler step, ``zero_grad``), for the configuration the published trainings use: frozen image encoder
(``freeze_backbone``), every other parameter learnable (``Lam.get_learnable_params``, models/lam.py:321-347): LAM neck, prompt
encoder, mask decoder, class encoder = 10.14 M parameters for the MAE-480 geometry.

``DecoderGraph`` restates the decoder side of ``Lam.forward`` (models/lam.py:115-136, prompt_encoder.py:564-827,
mask_decoder.py:316-363, lam.py:383-453) as a composition of ``autograd_ops`` - every arithmetic node runs a HIP kernel in both
directions, torch.autograd records the graph and moves data.  The image encoder runs through the inference engine without a graph.
Parameters that the forward never touches (``prompt_encoder.transformer.final_attn_token_to_image``, ``norm_final_attn``: dead in
the reference too, which therefore needs ``find_unused_parameters``) keep the zeros of the flat gradient buffer.
"""
from __future__ import annotations

import os

import math
from typing import Any, Dict, Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib as L
from . import autograd_ops as A
from .loss import FocalLossDevice
from .models import Lam
from .optim import FlatAdamW
from .parallel import any_over_ranks

Tensor = torch.Tensor


class _FocalObjective(Function):
    """LabelAnythingLoss({'focal': ...}, class_weighting) value with its hand-written gradient (la_focal_loss)."""

    @staticmethod
    def forward(ctx, logits, target, crit: FocalLossDevice):
        res = crit(logits.contiguous(), target, need_grad=True)
        ctx.save_for_backward(res["dlogits"])
        return res["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


class _PointEmbed(Function):
    """la_point_embed with gradients for the learned rows (point_embeddings.{0..3}, not_a_point_embed, no_sparse_embedding)."""

    @staticmethod
    def forward(ctx, type_emb, not_a_point, no_sparse, xy, kind, shift, d, image_size, gauss):
        out = xy.new_empty(kind.numel(), d)
        L.point_embed(xy, kind, shift, d, image_size, gauss, type_emb.contiguous(), not_a_point.contiguous(), no_sparse.contiguous(), out)
        ctx.save_for_backward(kind)
        return out

    @staticmethod
    def backward(ctx, g):
        (kind,) = ctx.saved_tensors
        k = kind.reshape(-1).long()
        g = g.reshape(k.numel(), -1)
        acc = g.new_zeros(6, g.shape[1]).index_add_(0, k, g)          # a handful of token rows: bookkeeping, not a kernel
        return acc[1:5].contiguous(), acc[0:1].contiguous(), acc[5:6].contiguous(), None, None, None, None, None, None


class _AddPerGroup(Function):
    """x [groups*rep, D] + y [groups, D] broadcast over the rep rows of its group (class encoding added to the (pair, hw) stream)."""

    @staticmethod
    def forward(ctx, x, y, groups, rep):
        d = x.shape[1]
        yb = x.new_empty(groups * rep, d)
        L.row_broadcast(y.contiguous(), groups, rep, d, 1.0, yb)
        out = torch.empty_like(yb)
        L.add_cast(x.contiguous(), yb, 0, out32=out, dt=L.LA_F32)
        ctx.dims = (groups, rep, d)
        return out

    @staticmethod
    def backward(ctx, g):
        groups, rep, d = ctx.dims
        dy = None
        if ctx.needs_input_grad[1]:
            dy = g.new_empty(groups, d)
            L.colmean(g.contiguous(), groups, rep, d, dy, g.new_empty(groups, L.COLMEAN_SPLIT, d))
            dy = dy * float(rep)
        return g, dy, None, None


class DecoderGraph:
    def __init__(self, lam: Lam, eng=None):
        self.lam = lam
        self.cfg = lam.cfg
        self.w: Dict[str, Tensor] = dict(lam.state_dict(keep_vars=True))
        self._pe: Dict[int, Tensor] = {}
        self.eng = eng if eng is not None else lam.engine()      # host-side helpers only (sparse-token layout)

    # ---- building blocks -----------------------------------------------------------------------------------------------------
    def lin(self, name: str, x: Tensor) -> Tensor:
        return A.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    def ln(self, name: str, x: Tensor, eps: float, gelu: bool = False) -> Tensor:
        return A.layer_norm(x, self.w[name + ".weight"], self.w[name + ".bias"], eps, gelu)

    def dense_pe(self, g: int) -> Tensor:
        t = self._pe.get(g)
        if t is None:
            d = self.cfg.embed_dim
            t = torch.empty(g * g, d, device=self.lam._device())
            L.dense_pe(self.w["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], g, d, t)
            self._pe[g] = t
        return t

    def conv_neck(self, pre: str, x: Tensor, bn: int, g: int) -> Tensor:
        """1x1 conv -> LN2d -> 3x3 conv -> LN2d on NHWC rows (build_lam.py:150-171)."""
        w0 = self.w[pre + ".0.weight"]
        a = A.linear(x, w0.reshape(w0.shape[0], -1))
        a = self.ln(pre + ".1", a, 1e-6)
        a = A.conv3x3(a, self.w[pre + ".2.weight"], None, bn, g, g)
        return self.ln(pre + ".3", a, 1e-6)

    def attn(self, pre: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, groups: int, nq: int, nk: int, q=None, k=None, v=None) -> Tensor:
        """Attention of models/common.py:57-148 (projections, softmax(qk^T/sqrt(c_head)) v, out_proj).  q / k / v: projections the caller
        already holds (two_way: the image-side ones of a layer come out of one ``A.linear_fan`` node)."""
        q = self.lin(pre + ".q_proj", q_in) if q is None else q
        k = self.lin(pre + ".k_proj", k_in) if k is None else k
        v = self.lin(pre + ".v_proj", v_in) if v is None else v
        return self.lin(pre + ".out_proj", A.attention(q, k, v, groups, nq, nk, self.cfg.dec_heads))

    def _wb(self, name: str, add_pe: bool):
        return (self.w[name + ".weight"], self.w.get(name + ".bias"), add_pe)

    def attention_mlp_block(self, pre: str, x: Tensor, groups: int, n: int) -> Tensor:
        """common.py:151-184: y = LN(attn(x) + x); out = LN(mlp(y) + y), one shared LayerNorm, GELU."""
        y = self.ln(pre + ".norm", A.add_rows(self.attn(pre + ".attn", x, x, x, groups, n, n), x), 1e-5)
        z = self.lin(pre + ".mlp.lin2", A.gelu(self.lin(pre + ".mlp.lin1", y)))
        return self.ln(pre + ".norm", A.add_rows(z, y), 1e-5)

    def two_way(self, pre: str, img: Tensor, pe: Tensor, tok: Tensor, groups: int, nt: int, hw: int, want_tokens: bool):
        """TwoWayTransformer (transformer.py:206-329): img [groups*hw, D], pe [hw, D] (no grad), tok [groups*nt, D]."""
        keys, tpe, qs = img, tok, tok
        for l in range(2):
            lp = f"{pre}.layers.{l}"
            if l == 0:
                qs = self.attn(lp + ".self_attn", qs, qs, qs, groups, nt, nt)                     # replaces the tokens
            else:
                qq = A.add_rows(qs, tpe)
                qs = A.add_rows(qs, self.attn(lp + ".self_attn", qq, qq, qs, groups, nt, nt))
            qs = self.ln(lp + ".norm1", qs, 1e-5)
            # the layer's three projections of the image-side stream - k (keys + pe), v (keys) of tokens -> image, q (keys + pe) of
            # image -> tokens - as ONE node: their data gradients meet in one buffer instead of two fan-in passes over the stream
            t2i, i2t = lp + ".cross_attn_token_to_image", lp + ".cross_attn_image_to_token"
            k_img, v_img, q_img = A.linear_fan(keys, pe, [self._wb(t2i + ".k_proj", True), self._wb(t2i + ".v_proj", False),
                                                          self._wb(i2t + ".q_proj", True)])
            qs = self.ln(lp + ".norm2", A.add_rows(qs, self.attn(t2i, A.add_rows(qs, tpe), None, None, groups, nt, hw, k=k_img, v=v_img)), 1e-5)
            m = self.lin(lp + ".mlp.lin2", A.relu(self.lin(lp + ".mlp.lin1", qs)))
            qs = self.ln(lp + ".norm3", A.add_rows(qs, m), 1e-5)
            keys = self.ln(lp + ".norm4", A.add_rows(keys, self.attn(i2t, None, A.add_rows(qs, tpe), qs, groups, hw, nt, q=q_img)), 1e-5)
        if not want_tokens:
            return None, keys
        fin = pre + ".final_attn_token_to_image"
        k_img, v_img = A.linear_fan(keys, pe, [self._wb(fin + ".k_proj", True), self._wb(fin + ".v_proj", False)])
        qs = self.ln(pre + ".norm_final_attn",
                     A.add_rows(qs, self.attn(fin, A.add_rows(qs, tpe), None, None, groups, nt, hw, k=k_img, v=v_img)), 1e-5)
        return qs, keys

    # ---- prompt encoder (prompt_encoder.py:564-827) ----------------------------------------------------------------------------
    def mask_embedding(self, masks: Tensor, flags: Tensor, p: int, g: int) -> Tensor:
        """mask_downscaling (conv2x2s2 -> LN2d -> GELU -> conv2x2s2 -> LN2d -> GELU -> conv1x1) as patch GEMMs, bilinear resize to the
        embedding grid, not_a_mask replacement (prompt_encoder.py:61-69,516-540,795-800) -> [P*hw, D]."""
        pre = "prompt_encoder.mask_downscaling"
        d = self.cfg.embed_dim
        hm = masks.shape[-1]
        h2, h4 = hm // 2, hm // 4
        x = masks.reshape(p, h2, 2, h2, 2).permute(0, 1, 3, 2, 4).reshape(p * h2 * h2, 4)          # non-overlapping 2x2 patches
        w0 = self.w[pre + ".0.weight"]
        a = A.linear(x.contiguous(), w0.reshape(w0.shape[0], 4), self.w[pre + ".0.bias"])
        a = self.ln(pre + ".1", a, 1e-6, gelu=True)
        c1 = a.shape[1]
        a = a.reshape(p, h4, 2, h4, 2, c1).permute(0, 1, 3, 2, 4, 5).reshape(p * h4 * h4, 4 * c1)   # (ky, kx, cin)
        w3 = self.w[pre + ".3.weight"]
        a = A.linear(a.contiguous(), w3.permute(0, 2, 3, 1).reshape(w3.shape[0], -1), self.w[pre + ".3.bias"])
        a = self.ln(pre + ".4", a, 1e-6, gelu=True)
        w6 = self.w[pre + ".6.weight"]
        dense = A.linear(a, w6.reshape(w6.shape[0], -1), self.w[pre + ".6.bias"])                    # [P*h4*h4, D]
        if h4 != g:
            if d % 4 == 0:
                dense = A.bilinear_rows(dense, p, h4, h4, d, g, g)           # on the NHWC rows: no plane transposes around the resize
            else:
                planes = A.rows_to_planes(dense, p, d, h4 * h4).reshape(p * d, h4, h4)
                planes = A.bilinear(planes, g, g)
                dense = A.planes_to_rows(planes.reshape(p * d, g * g), p, d, g * g)
        missing = (flags.reshape(p) == 0).view(p, 1, 1)
        nam = self.w["prompt_encoder.not_a_mask_embed.weight"].view(1, 1, d)
        return torch.where(missing, nam, dense.reshape(p, g * g, d)).reshape(p * g * g, d)

    def sparse_tokens(self, b: int, m: int, c: int, points, boxes) -> Tuple[Tensor, int]:
        eng = self.eng
        (xy, kind, shift), ns = eng._sparse_tokens(b, m, c, points, boxes)
        pe_ = "prompt_encoder"
        # only the rows a prompt type actually uses enter the graph: the reference never touches point_embeddings.2/3 without boxes,
        # .0/.1 without points, no_sparse_embedding with any sparse prompt (their .grad stays None and AdamW skips them)
        used = [points is not None, points is not None, boxes is not None, boxes is not None]
        type_emb = torch.cat([self.w[f"{pe_}.point_embeddings.{i}.weight"] if used[i] else self.w[f"{pe_}.point_embeddings.{i}.weight"].detach()
                              for i in range(4)])
        any_sparse = points is not None or boxes is not None
        nap = self.w[pe_ + ".not_a_point_embed.weight"]
        nsp = self.w[pe_ + ".no_sparse_embedding.weight"]
        sp = _PointEmbed.apply(type_emb, nap if any_sparse else nap.detach(), nsp.detach() if any_sparse else nsp, xy, kind,
                               shift, self.cfg.embed_dim, self.cfg.image_size,
                               self.w[pe_ + ".pe_layer.positional_encoding_gaussian_matrix"])
        return sp, ns

    def prompt_encoder(self, support: Tensor, b: int, m: int, g: int, points, boxes, masks, flag_examples: Tensor,
                       selected_rows: Optional[Tensor]) -> Dict[str, Tensor]:
        cfg = self.cfg
        pe_ = "prompt_encoder"
        d, hw = cfg.embed_dim, g * g
        first = points[0] if points is not None else boxes[0] if boxes is not None else masks[0] if masks is not None else None
        if first is None:
            raise ValueError("No prompts provided")
        c = first.shape[2]
        p = b * m * c
        sp, ns = self.sparse_tokens(b, m, c, points, boxes)
        sp = self.attention_mlp_block(pe_ + ".sparse_embedding_attention", sp, b * m, c * ns)
        ce = None
        if cfg.bank_size:
            ce = self.w[pe_ + ".class_encoder.pos_embedding"][0, 0].index_select(0, selected_rows)         # (C, D)
            sp = A.add_rows(sp, ce.repeat_interleave(ns, dim=0))                                            # rows ordered (c, n)
        if masks is not None:
            mk, mf = masks
            dense = self.mask_embedding(mk.reshape(p, mk.shape[-2], mk.shape[-1]), mf, p, g)
        else:
            dense = self.w[pe_ + ".no_mask_embed.weight"].view(1, d).expand(p * hw, d)
        sup = support.view(b * m, 1, hw, d).expand(b * m, c, hw, d).reshape(p * hw, d)                      # every class sees its support
        src = A.add_rows(sup, dense.contiguous())
        if ce is not None:
            src = _AddPerGroup.apply(src, ce.repeat(b * m, 1), p, hw)
        pos = self.dense_pe(g)
        _, keys = self.two_way(pe_ + ".transformer", src, pos, sp, p, ns, hw, want_tokens=False)
        emb = A.mean_rows(keys, p, hw)
        if cfg.class_attention:
            emb = self.attention_mlp_block(pe_ + ".class_attention", emb, b * m, c)
        if cfg.example_attention:
            e2 = emb.view(b, m, c, d).permute(0, 2, 1, 3).reshape(b * c * m, d)
            e2 = self.attention_mlp_block(pe_ + ".example_attention", e2, b * c, m)
            emb = e2.view(b, c, m, d).permute(0, 2, 1, 3).reshape(p, d)
        if cfg.example_class_attention:
            emb = self.attention_mlp_block(pe_ + ".class_example_attention", emb, b, m * c)
        # class prototypes: masked mean over the supports (prompt_encoder.py:738-745); (B, M, C) bookkeeping on a few rows
        fe = flag_examples.reshape(b, m, c).to(emb.dtype).unsqueeze(-1)
        denom = fe.sum(dim=1)
        denom = torch.where(denom == 0, torch.ones_like(denom), denom)
        cls = (emb.view(b, m, c, d) * fe).sum(dim=1) / denom
        return {"class_embeddings": cls, "class_examples_embeddings": emb.view(b, m, c, d), "class_examples_src": keys}

    # ---- mask decoder (mask_decoder.py:316-363) --------------------------------------------------------------------------------
    def conv_transpose_2x2(self, name: str, x: Tensor, bsz: int, h: int, wd: int) -> Tensor:
        """ConvTranspose2d(k=2, s=2) as a GEMM + pixel shuffle: rows (b, y, x) -> rows (b, 2y+ky, 2x+kx)."""
        w = self.w[name + ".weight"]                                    # (Cin, Cout, 2, 2)
        cout = w.shape[1]
        y = A.linear(x, w.permute(2, 3, 1, 0).reshape(4 * cout, w.shape[0]), self.w[name + ".bias"].repeat(4))
        return y.view(bsz, h, wd, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(bsz * 4 * h * wd, cout)

    def mask_decoder(self, query: Tensor, b: int, g: int, class_emb: Tensor) -> Tensor:
        cfg = self.cfg
        md = "mask_decoder"
        d, hw = cfg.embed_dim, g * g
        c = class_emb.shape[1]
        toks, keys = self.two_way(md + ".transformer", query, self.dense_pe(g), class_emb.reshape(b * c, d), b, c, hw, want_tokens=True)
        pr = A.relu(self.lin(md + ".class_mlp.layers.0", toks))
        pr = A.relu(self.lin(md + ".class_mlp.layers.1", pr))
        pr = self.lin(md + ".class_mlp.layers.2", pr)
        up = self.conv_transpose_2x2(md + ".output_upscaling.0", keys, b, g, g)
        up = self.ln(md + ".output_upscaling.1", up.contiguous(), 1e-6, gelu=True)
        feat = self.conv_transpose_2x2(md + ".output_upscaling.3", up, b, 2 * g, 2 * g).contiguous()
        if cfg.spatial_convs:
            for i in range(cfg.spatial_convs):
                feat = A.conv3x3(feat, self.w[f"{md}.spatial_convs.{3 * i}.weight"], self.w[f"{md}.spatial_convs.{3 * i}.bias"], b, 4 * g, 4 * g)
                if i < cfg.spatial_convs - 1:
                    feat = self.ln(f"{md}.spatial_convs.{3 * i + 1}", feat, 1e-6, gelu=True)
        seg = A.classify(feat, pr.reshape(b, c, -1), b, 16 * hw, c)
        return seg.view(b, c, 4 * g, 4 * g)

    # ---- post-processing (lam.py:383-453) --------------------------------------------------------------------------------------
    def postprocess(self, seg: Tensor, dims: Tensor, flag_gts: Optional[Tensor]) -> Tensor:
        cfg = self.cfg
        s = cfg.image_size
        b, c, h, w = seg.shape
        big = A.bilinear(seg.reshape(b * c, h, w), s, s).view(b, c, s, s)
        dl = dims.detach().to("cpu").tolist()
        hmax = max(int(x[0]) for item in dl for x in item)
        wmax = max(int(x[1]) for item in dl for x in item)
        outs = []
        for i, item in enumerate(dl):
            oh, ow = int(item[0][0]), int(item[0][1])
            one = big[i]
            if cfg.custom_preprocess:
                sc = s * 1.0 / max(oh, ow)
                one = one[:, :int(oh * sc + 0.5), :int(ow * sc + 0.5)]
            if one.shape[-2:] != (oh, ow):
                one = A.bilinear(one.contiguous(), oh, ow)
            pad = one.new_full((c, hmax, wmax), float("-inf"))
            pad[0] = 0.0                                              # the background class is padded with 0 (lam.py:444-446)
            pad[:, :oh, :ow] = one
            outs.append(pad)
        out = torch.stack(outs)
        if flag_gts is not None:
            out = out.masked_fill(flag_gts.to(out.device).logical_not().view(b, c, 1, 1), float("-inf"))
        return out

    # ---- whole decoder side of Lam.forward --------------------------------------------------------------------------------------
    def forward(self, e_rows: Tensor, b: int, n: int, g: int, inp: Dict[str, Tensor], dims: Tensor, neck_input: bool) -> Dict[str, Tensor]:
        """e_rows: [B*N*hw, C] NHWC rows out of the (frozen) encoder or of precomputed embeddings (no grad)."""
        cfg = self.cfg
        d, hw = cfg.embed_dim, g * g
        if cfg.lam_neck and neck_input:
            e_rows = self.conv_neck("neck", e_rows, b * n, g)
        ev = e_rows.view(b, n, hw, d)
        query = ev[:, 0].reshape(b * hw, d)
        support = ev[:, 1:].reshape(b * (n - 1) * hw, d)
        points, boxes, masks = Lam._prompts_of(inp)
        pe = self.prompt_encoder(support, b, n - 1, g, points, boxes, masks, inp["flag_examples"], inp.get("selected_rows"))
        seg = self.mask_decoder(query.contiguous(), b, g, pe["class_embeddings"])
        logits = self.postprocess(seg, dims, inp.get("flag_gts"))
        return {"logits": logits, "low_res_logits": seg, **pe}


class LamTrainer:
    """One data-parallel training replica: ``step(batch, gt)`` = forward + focal objective + backward + (RCCL) gradient all-reduce +
    AdamW with HF's constant-with-warm-up schedule (experiment/utils.py:53-100), on the learnable parameters of
    ``Lam.get_learnable_params({'freeze_backbone': True})``."""

    def __init__(self, lam: Lam, lr: float = 5e-5, weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8,
                 num_warmup_steps: int = 0, loss: Optional[FocalLossDevice] = None, train_encoder: bool = False,
                 backbone_lr: Optional[float] = None, encoder_buckets: int = 4):
        """backbone_lr: learning rate of the ``image_encoder.*`` tensors (the reference's ``backbone_lr`` parameter group,
        models/lam.py:340-346; needs train_encoder=True - with a frozen backbone the reference raises as well).
        train_encoder=False: ``get_learnable_params({'freeze_backbone': True})`` - the image encoder is frozen (and absent from
        the flat buffer).  encoder_buckets: the encoder's share of the flat gradient (345 MB for ViT-B) is all-reduced in this many
        pieces, each followed by its AdamW launch (``parallel.BucketedGradReducer``); the decoder-side gradients (40 MB) are one more
        bucket, launched from inside the backward pass - they are final when the encoder backward starts.
        train_encoder=True: every parameter trains, as with parameters/trainval/coco20i/mae_noembs.yaml (no
        ``freeze_backbone``: models/lam.py:347 returns ``self.parameters()``); needs an HF ViT encoder (train_encoder.py)."""
        if float(getattr(lam.cfg, "dropout", 0.0) or 0.0) != 0.0:
            # the reference trains with nn.Dropout(p) inside MLPBlock / AttentionMLPBlock when it is set (models/common.py:25-32,68-75,
            # build_lam.py:128); the training graph here has no dropout node, so training such a model would silently be a different model
            raise NotImplementedError(f"dropout={lam.cfg.dropout} is not built into the training graph (every canonical parameters/*.yaml "
                                      f"leaves it at 0); build the model with dropout=0.0 to train it here")
        if lam._device().type != "cuda":
            raise RuntimeError("LamTrainer needs the model on an MI355X (there is no CPU path)")
        self.lam = lam
        self.train_encoder = bool(train_encoder)
        if self.train_encoder and lam.cfg.encoder_spec is None:
            raise ValueError("train_encoder=True needs a model with an image encoder")
        if backbone_lr is not None and not self.train_encoder:
            raise ValueError("Cannot freeze the backbone and set a learning rate for it at the same time.")
        # the saved-activation forward of train_encoder.py runs the plain GEMM sequence, without the inference engine's token-mean
        # corrections of V / proj: what those recover (logits 9.5e-4 -> 6e-4) is far below the 16-bit backward's own error (1e-3 ... 1e-2
        # on the gradients), and second weight planes in their place cost 2.5 ms of the step.  That choice lives in the
        # trainer's PRIVATE encoder engine (HfEncoderGraph): ``lam.precise`` - what validation / inference on the same model use - is
        # left alone.
        self.train_precise = tuple(gname for gname in lam.precise if gname not in ("vmean", "projmean"))
        named = [(k, p) for k, p in lam.named_parameters() if self.train_encoder or "image_encoder" not in k]
        # tensors the forward never reaches (dead in the reference too, prompt_encoder.py:683) go to the tail of the flat buffer so
        # that the per-step "received a gradient" spans of FlatAdamW.step stay one contiguous run
        dead = ("prompt_encoder.transformer.final_attn_token_to_image.", "prompt_encoder.transformer.norm_final_attn.")
        named = [kp for kp in named if not kp[0].startswith(dead)] + [kp for kp in named if kp[0].startswith(dead)]
        for _, p in lam.named_parameters():
            p.requires_grad_(False)
        for _, p in named:
            p.requires_grad_(True)
        self.names = [k for k, _ in named]
        # "received a gradient since the last zero_grad()": OR-ed over the micro-steps of a gradient accumulation (the reference
        # accumulates over substitution steps, experiment/run.py:503-527) and, in apply_update, over the ranks
        self._touched = [False] * len(named)
        for i, (_, p) in enumerate(named):
            p.register_post_accumulate_grad_hook(lambda _p, _i=i: self._on_grad(_i))
        lrs = None if backbone_lr is None else [backbone_lr if "image_encoder" in k else lr for k, _ in named]
        self.opt = FlatAdamW([p for _, p in named], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                             num_warmup_steps=num_warmup_steps, lrs=lrs)
        for (_, p), gv in zip(named, self.opt.grad_views):
            p.grad = gv                      # autograd accumulates straight into the flat gradient buffer
        # gradient buckets in buffer order: the encoder tensors (first in named_parameters order) in ``encoder_buckets`` pieces cut at
        # tensor boundaries, everything else (neck, prompt encoder, mask decoder, dead tail) as the last one
        sizes = [p.numel() for _, p in named]
        n_enc = sum(1 for k, _ in named if k.startswith("image_encoder."))
        if any(k.startswith("image_encoder.") for k, _ in named[n_enc:]):
            raise RuntimeError("image_encoder tensors are expected to lead the parameter order")
        enc_total, bounds, off, cut = sum(sizes[:n_enc]), [], 0, 0
        nb = max(1, int(encoder_buckets))
        for i in range(n_enc):
            off += sizes[i]
            if off - cut >= enc_total / nb and off < enc_total:
                bounds.append((cut, off))
                cut = off
        if enc_total > cut:
            bounds.append((cut, enc_total))
        self._enc_buckets = list(range(len(bounds)))
        self._dec_bucket = len(bounds)
        bounds.append((enc_total, sum(sizes)))
        from .parallel import BucketedGradReducer
        self.reducer = BucketedGradReducer(self.opt.grad, bounds)
        self._sink = A.GradSink([p for _, p in named], self.opt.grad_views, self._on_grad)
        self._wt = A.WeightTransposes(enabled=True, lo=self.opt.flat.data_ptr(), hi=self.opt.flat.data_ptr() + 4 * self.opt.flat.numel())
        self._dec_index0 = n_enc                 # parameters [n_enc, ...) live in the decoder bucket
        self.crit = loss or FocalLossDevice()
        self.engine = lam.engine()           # host-side helpers + the frozen encoder (its packed weights never change)
        self.graph = DecoderGraph(lam, self.engine)
        self._seen_version = lam.weights_version      # out-of-band weight changes (load_state_dict, invalidate) re-derive both: _sync_version
        self.enc_graph = None
        if self.train_encoder:
            from .train_encoder import HfEncoderGraph, SamEncoderGraph
            graph_cls = SamEncoderGraph if lam.cfg.encoder_spec.kind == "sam" else HfEncoderGraph
            self.enc_graph = graph_cls(lam, {k: gv for k, gv in zip(self.names, self.opt.grad_views) if k.startswith("image_encoder.")},
                                       precise=self.train_precise)
            self._anchor = torch.zeros(1, device=lam._device(), requires_grad=True)
            self._enc_idx = [i for i, k in enumerate(self.names) if graph_cls.owns(k)]

    def _sync_version(self) -> None:
        """A checkpoint restore / manual edit + ``lam.invalidate()`` since the last call: everything this trainer derived from the weights
        is rebuilt - its engine (frozen-encoder packs, host-side helpers) and the decoder graph's cached position table (from the
        ``positional_encoding_gaussian_matrix`` buffer); the encoder graph follows ``lam.weights_version`` itself.  The trainer's own
        optimizer steps do not count: they move neither the frozen encoder nor the buffers."""
        if self.lam.weights_version != self._seen_version:
            self.engine = self.lam.engine()
            self.graph.eng = self.engine
            self.graph._pe.clear()
            self._seen_version = self.lam.weights_version

    def _on_grad(self, i: int) -> None:
        self._touched[i] = True
        if i >= self._dec_index0:
            self.reducer.invalidate(self._dec_bucket)      # (only matters if the bucket's staged reduction has already been launched)

    def forward_backward(self, batch: Dict[str, Any], gt: Tensor, loss_normalizer: float = 1.0, sync: bool = False) -> Dict[str, Tensor]:
        """sync=True: this is the LAST micro-step before ``apply_update`` (DDP's "not no_sync"): gradient buckets are handed to the
        all-reduce as they become final - the decoder-side bucket when the encoder backward starts, the encoder buckets right behind
        the backward pass - and ``apply_update`` must be the next call.  sync=False (default): gradients only accumulate; whatever has
        not been reduced by then is reduced inside ``apply_update``."""
        lam = self.lam
        if any(self.reducer.launched(i) for i in range(len(self.reducer.bounds))):
            raise RuntimeError("forward_backward after a synchronising micro-step: call apply_update() first")
        self._sync_version()
        prev_wt, prev_sink = A.WT, A.SINK
        A.WT = self._wt              # the W^T copies of this trainer's weights (flat-buffer views: stable addresses), one launch per step
        self._wt.invalidate()
        A.SINK = self._sink          # nn.Linear / LayerNorm parameter gradients are added straight into the flat gradient buffer
        try:
            return self._forward_backward(batch, gt, loss_normalizer, sync)
        finally:                     # a backward through these operators outside this call must not see this trainer's W^T copies / sink
            A.WT, A.SINK = prev_wt, prev_sink

    def _forward_backward(self, batch, gt, loss_normalizer, sync):
        lam = self.lam
        with torch.cuda.device(lam._device()):
            with torch.no_grad():
                inp, _ = lam._prepare(batch, with_post=False, eng=self.engine)
                if "flag_gts" in batch:
                    inp["flag_gts"] = batch["flag_gts"].to(lam._device())
                eng = self.engine
                through_encoder = False
                if "embeddings" in inp:
                    emb = inp["embeddings"]
                    b, n, c, h, w = emb.shape
                    g = h
                    e_rows = torch.empty(b * n * h * w, c, device=emb.device)
                    L.nchw_to_nhwc(emb.reshape(b * n, c, h * w).contiguous(), b * n, c, h * w, out32=e_rows, dt=L.LA_F32)
                else:
                    im = inp["images"]
                    b, n = im.shape[:2]
                    g = im.shape[-1] // lam.cfg.encoder_spec.patch
                    if self.enc_graph is None:
                        e32, _, c, g = eng.encode_images(im.flatten(0, 1))
                        e_rows = e32.clone()
                    else:
                        through_encoder = True
            if through_encoder:                       # forward with saved activations; its backward hangs off the decoder graph's
                from .train_encoder import encode_trainable
                e_rows = encode_trainable(self.enc_graph, im.flatten(0, 1), self._anchor)
                for i in self._enc_idx:
                    self._touched[i] = True
                if lam.cfg.encoder_spec.kind == "sam" and lam.cfg.use_vit_sam_neck:
                    # the SAM neck (image_encoder.py:92-108) on the decoder graph's autograd operators, exact fp32
                    e_rows = self.graph.conv_neck("image_encoder.neck", e_rows, b * n, g)
                # the encoder's backward node is the last one autograd runs (it was created first): every decoder-side gradient has
                # been accumulated by then, so their bucket starts travelling under the encoder backward (a copy is reduced; a
                # gradient that does arrive later invalidates it and the bucket is reduced in place at the end)
                self.enc_graph.before_backward = (lambda: self.reducer.launch(self._dec_bucket, staged=True)) if sync else None
            out = self.graph.forward(e_rows, b, n, g, inp, batch["dims"], neck_input=True)
            loss = _FocalObjective.apply(out["logits"], gt.to(lam._device()), self.crit)
            (loss / loss_normalizer).backward()
            if sync:                                  # everything is final now: the remaining buckets leave in buffer order
                for i in range(len(self.reducer.bounds)):
                    if not self.reducer.launched(i):
                        self.reducer.launch(i)
        return {"loss": loss.detach(), "logits": out["logits"].detach(), "class_examples_embeddings": out["class_examples_embeddings"].detach()}

    def zero_grad(self) -> None:
        self.opt.zero_grad()
        self._touched = [False] * len(self.names)
        self.reducer.begin()

    def apply_update(self) -> None:
        """SUM all-reduce of the flat gradient over the ranks + the AdamW launches over the tensors that received a gradient ON ANY
        RANK since the last ``zero_grad``.  Prompt types are sampled per episode (data/dataset.py:292), so one rank can see points
        and another none: the flags are OR-ed over the ranks first (DDP ``find_unused_parameters=True`` all-reduces its
        used-parameter bitmap the same way, experiment/run.py:123) - otherwise replicas of ``point_embeddings.*``,
        ``not_a_point_embed``, ``mask_downscaling.*`` ... would step on some ranks only and drift apart."""
        # one exchange: used-parameter flags + "my staged bucket went stale" flags.  The stale decision must be COLLECTIVE: a rank that
        # re-reduces a bucket in place issues an all-reduce its peers would never issue (mis-paired collectives / a hang)
        nb = len(self.reducer.bounds)
        flags = any_over_ranks(list(self._touched) + self.reducer.stale_flags(), device=self.lam._device())
        active = flags[:len(self._touched)]
        self.reducer.set_stale(flags[len(self._touched):len(self._touched) + nb])
        self.opt.step(active=active, reducer=self.reducer)
        self._wt.invalidate()
        self.reducer.begin()
        self._touched = [False] * len(self.names)
        # packed / converted weight copies of the inference engine are stale now; the trainer's own engine only serves the frozen
        # encoder and host-side helpers, so nothing of it is re-packed until the model is next used for inference; a trainable
        # encoder's private engine re-packs its encoder weights (and only those) before the next forward
        external = self.lam.weights_version != self._seen_version      # (a restore between forward_backward and here: keep it pending)
        self.lam.invalidate()
        if not external:
            self._seen_version = self.lam.weights_version
        if self.enc_graph is not None:
            self.enc_graph.weights_changed()

    def step(self, batch: Dict[str, Any], gt: Tensor, loss_normalizer: float = 1.0) -> Dict[str, Tensor]:
        self.zero_grad()
        res = self.forward_backward(batch, gt, loss_normalizer, sync=True)
        self.apply_update()
        return res
