"""CPU oracle for the LabelAnything hot path (TEST INFRASTRUCTURE ONLY).

This file is a from-the-math fp32 restatement, in plain functional torch on CPU,
of the reference's inference path: ViT image encoder (SAM ViTDet flavour and the
HuggingFace plain-ViT flavour), the LAM neck, the prompt encoder (mask / point /
box prompts, two-way token<->image transformer, class prototypes), the mask
decoder (two-way transformer, transposed-conv upscaler, spatial convs, prototype
classification) and logit post-processing.

It is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (labelanything_amd) never does.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md 4,
8c), so this oracle is pinned against outputs of the reference itself imported
in the build container (tools/make_golden.py -> tests/golden/*.safetensors).

All weights are passed as a flat ``dict[str, Tensor]`` keyed exactly like the
reference's ``Lam.state_dict()`` (no ``model.`` prefix), e.g.
``image_encoder.blocks.3.attn.qkv.weight``.  Reference citations are
``path:line`` under /root/reference/label_anything/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
@dataclass
class LamGeometry:
    """Model geometry (the ``model:`` section of a reference YAML / LabelAnything kwargs).

    encoder: "sam" (models/image_encoder.py), "hf" (transformers ViTModel via
    models/build_encoder.py:83-100) or None (precomputed embeddings, lam_no_vit).
    """

    encoder: Optional[str] = "sam"
    image_size: int = 1024
    patch: int = 16
    enc_dim: int = 768
    enc_depth: int = 12
    enc_heads: int = 12
    enc_mlp: int = 3072
    global_idx: Tuple[int, ...] = (2, 5, 8, 11)
    window: int = 14
    sam_neck: bool = True          # use_vit_sam_neck / project_last_hidden
    sam_out: int = 256             # out_chans of the SAM neck
    hf_pos_grid: int = 14          # sqrt(num_positions) of the HF checkpoint (224/16)
    image_embed_dim: int = 256     # channels entering the LAM neck / decoder
    embed_dim: int = 256           # D
    class_attention: bool = False
    example_attention: bool = False
    example_class_attention: bool = True
    class_encoder_bank: int = 0    # 0 = no RandomMatrixEncoder
    spatial_convs: Optional[int] = 3
    custom_preprocess: bool = True
    dec_heads: int = 8
    dec_mlp: int = 2048
    mask_in_chans: int = 16

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def lam_neck(self) -> bool:
        return self.image_embed_dim != self.embed_dim


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def linear(w: W, name: str, x: Tensor) -> Tensor:
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def layer_norm(w: W, name: str, x: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], eps)


def layer_norm_2d(w: W, name: str, x: Tensor, eps: float = 1e-6) -> Tensor:
    """Per-pixel normalisation over channels of an NCHW map, biased variance (models/common.py:42-54)."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    xn = (x - mu) / torch.sqrt(var + eps)
    return xn * w[name + ".weight"].view(1, -1, 1, 1) + w[name + ".bias"].view(1, -1, 1, 1)


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------------------
# SAM ViTDet encoder  (models/image_encoder.py)
# --------------------------------------------------------------------------------------
def rel_pos_table(size_q: int, size_k: int, table: Tensor) -> Tensor:
    """R[i, j] = table[(i - j) + (size_k - 1)] for equal sizes; general form per image_encoder.py:307-337."""
    span = 2 * max(size_q, size_k) - 1
    if table.shape[0] != span:
        t = F.interpolate(table.t().unsqueeze(0), size=span, mode="linear")[0].t()
    else:
        t = table
    qi = torch.arange(size_q, dtype=torch.float32).view(-1, 1) * max(size_k / size_q, 1.0)
    kj = torch.arange(size_k, dtype=torch.float32).view(1, -1) * max(size_q / size_k, 1.0)
    idx = (qi - kj + (size_k - 1) * max(size_q / size_k, 1.0)).long()
    return t[idx]  # (size_q, size_k, hd)


def sam_attention(w: W, pre: str, x: Tensor, heads: int) -> Tensor:
    """x: (N, h, w, E) tokens of one attention domain (whole grid or one window).  image_encoder.py:200-255."""
    n, gh, gw, e = x.shape
    hd = e // heads
    t = gh * gw
    qkv = linear(w, pre + ".qkv", x.reshape(n, t, e))           # rows ordered q|k|v, head-major inside each
    qkv = qkv.view(n, t, 3, heads, hd).permute(2, 0, 3, 1, 4)   # (3, n, heads, t, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scores = (q * (hd ** -0.5)) @ k.transpose(-1, -2)           # scaled q for the content term
    if (pre + ".rel_pos_h") in w:
        rh = rel_pos_table(gh, gh, w[pre + ".rel_pos_h"])       # (gh, gh, hd)
        rw = rel_pos_table(gw, gw, w[pre + ".rel_pos_w"])
        qg = q.reshape(n, heads, gh, gw, hd)                    # UNSCALED q for the positional terms (:246-249)
        bias_h = torch.einsum("nhyxc,ykc->nhyxk", qg, rh)       # (n, heads, gh, gw, kh)
        bias_w = torch.einsum("nhyxc,xkc->nhyxk", qg, rw)       # (n, heads, gh, gw, kw)
        scores = scores.view(n, heads, gh, gw, gh, gw) + bias_h[..., :, None] + bias_w[..., None, :]
        scores = scores.view(n, heads, t, t)
    p = torch.softmax(scores, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(n, gh, gw, e)
    return linear(w, pre + ".proj", o)


def window_split(x: Tensor, ws: int) -> Tuple[Tensor, Tuple[int, int]]:
    """(B,H,W,C) -> (B*nWin, ws, ws, C) with zero padding bottom/right (image_encoder.py:258-279)."""
    b, h, wd, c = x.shape
    ph, pw = (-h) % ws, (-wd) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    hp, wp = h + ph, wd + pw
    x = x.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, c), (hp, wp)


def window_merge(xw: Tensor, ws: int, padded: Tuple[int, int], hw: Tuple[int, int]) -> Tensor:
    hp, wp = padded
    h, wd = hw
    b = xw.shape[0] // ((hp // ws) * (wp // ws))
    x = xw.view(b, hp // ws, wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, -1)
    return x[:, :h, :wd, :]


def sam_block(w: W, pre: str, x: Tensor, heads: int, window: int) -> Tensor:
    """image_encoder.py:179-197; LN eps 1e-6 (build_encoder.py:61); padding happens AFTER norm1."""
    y = layer_norm(w, pre + ".norm1", x, 1e-6)
    if window > 0:
        h, wd = y.shape[1], y.shape[2]
        y, padded = window_split(y, window)
        y = sam_attention(w, pre + ".attn", y, heads)
        y = window_merge(y, window, padded, (h, wd))
    else:
        y = sam_attention(w, pre + ".attn", y, heads)
    x = x + y
    z = layer_norm(w, pre + ".norm2", x, 1e-6)
    z = linear(w, pre + ".mlp.lin2", gelu(linear(w, pre + ".mlp.lin1", z)))
    return x + z


def conv_neck(w: W, pre: str, x: Tensor) -> Tensor:
    """1x1 conv -> LN2d -> 3x3 conv (pad 1) -> LN2d, both convs bias-free.
    SAM neck image_encoder.py:92-108 and the LAM neck build_lam.py:150-171 share this shape."""
    x = F.conv2d(x, w[pre + ".0.weight"])
    x = layer_norm_2d(w, pre + ".1", x)
    x = F.conv2d(x, w[pre + ".2.weight"], padding=1)
    return layer_norm_2d(w, pre + ".3", x)


def sam_encoder(w: W, geo: LamGeometry, images: Tensor, pre: str = "image_encoder",
                return_last_block: bool = False):
    """images (Bn,3,S,S) -> (Bn, sam_out, g, g) (or (Bn,E,g,g) without the neck).  image_encoder.py:110-131."""
    x = F.conv2d(images, w[pre + ".patch_embed.proj.weight"], w[pre + ".patch_embed.proj.bias"],
                 stride=geo.patch).permute(0, 2, 3, 1)
    if (pre + ".pos_embed") in w:
        x = x + w[pre + ".pos_embed"]
    for i in range(geo.enc_depth):
        win = 0 if i in geo.global_idx else geo.window
        x = sam_block(w, f"{pre}.blocks.{i}", x, geo.enc_heads, win)
    last = x.permute(0, 3, 1, 2)
    if not geo.sam_neck:
        return last
    out = conv_neck(w, pre + ".neck", last)
    return (out, last.clone()) if return_last_block else out


# --------------------------------------------------------------------------------------
# HuggingFace plain ViT encoder (transformers ViTModel maths, 4.x state-dict names)
#   call sites: models/build_encoder.py:83-100, preprocess.py:193-206
# --------------------------------------------------------------------------------------
def hf_pos_embed(w: W, pre: str, g: int, pos_grid: int) -> Tensor:
    """CLS position kept, patch positions bicubic-resampled pos_grid^2 -> g^2 (align_corners=False)."""
    pos = w[pre + ".embeddings.position_embeddings"]            # (1, 1+pos_grid^2, E)
    if g == pos_grid:
        return pos
    e = pos.shape[-1]
    grid = pos[:, 1:].reshape(1, pos_grid, pos_grid, e).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(g, g), mode="bicubic", align_corners=False)
    return torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, g * g, e)], dim=1)


def hf_vit_encoder(w: W, geo: LamGeometry, images: Tensor, pre: str = "image_encoder") -> Tensor:
    """images (Bn,3,S,S) -> (Bn,E,g,g): pre-LN blocks with CLS kept to the end, final LN, CLS dropped."""
    bn = images.shape[0]
    g = images.shape[-1] // geo.patch
    e, heads = geo.enc_dim, geo.enc_heads
    hd = e // heads
    x = F.conv2d(images, w[pre + ".embeddings.patch_embeddings.projection.weight"],
                 w[pre + ".embeddings.patch_embeddings.projection.bias"], stride=geo.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([w[pre + ".embeddings.cls_token"].expand(bn, -1, -1), x], dim=1)
    x = x + hf_pos_embed(w, pre, g, geo.hf_pos_grid)
    t = x.shape[1]
    eps = 1e-12
    for i in range(geo.enc_depth):
        lp = f"{pre}.encoder.layer.{i}"
        y = layer_norm(w, lp + ".layernorm_before", x, eps)
        q = linear(w, lp + ".attention.attention.query", y).view(bn, t, heads, hd).transpose(1, 2)
        k = linear(w, lp + ".attention.attention.key", y).view(bn, t, heads, hd).transpose(1, 2)
        v = linear(w, lp + ".attention.attention.value", y).view(bn, t, heads, hd).transpose(1, 2)
        p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (p @ v).transpose(1, 2).reshape(bn, t, e)
        x = x + linear(w, lp + ".attention.output.dense", o)
        y = layer_norm(w, lp + ".layernorm_after", x, eps)
        y = gelu(linear(w, lp + ".intermediate.dense", y))
        x = x + linear(w, lp + ".output.dense", y)
    x = layer_norm(w, pre + ".layernorm", x, eps)
    return x[:, 1:, :].reshape(bn, g, g, e).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# decoder-side attention primitives  (models/common.py, models/transformer.py)
# --------------------------------------------------------------------------------------
def dec_attention(w: W, pre: str, q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """common.py:57-148.  Scores are divided by sqrt(c_per_head) AFTER the matmul; key_mask is a no-op."""
    q = linear(w, pre + ".q_proj", q)
    k = linear(w, pre + ".k_proj", k)
    v = linear(w, pre + ".v_proj", v)
    b, nq, ci = q.shape
    ch = ci // heads
    qh = q.view(b, nq, heads, ch).transpose(1, 2)
    kh = k.view(b, k.shape[1], heads, ch).transpose(1, 2)
    vh = v.view(b, v.shape[1], heads, ch).transpose(1, 2)
    a = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(ch), dim=-1)
    o = (a @ vh).transpose(1, 2).reshape(b, nq, ci)
    return linear(w, pre + ".out_proj", o)


def attention_mlp_block(w: W, pre: str, x: Tensor, heads: int) -> Tensor:
    """common.py:151-184: one shared LayerNorm (eps 1e-5) used twice; GELU MLP."""
    y = layer_norm(w, pre + ".norm", dec_attention(w, pre + ".attn", x, x, x, heads) + x, 1e-5)
    z = linear(w, pre + ".mlp.lin2", gelu(linear(w, pre + ".mlp.lin1", y)))
    return layer_norm(w, pre + ".norm", z + y, 1e-5)


def two_way_transformer(w: W, pre: str, image: Tensor, image_pe: Tensor, tokens: Tensor,
                        heads: int, depth: int = 2) -> Tuple[Tensor, Tensor]:
    """transformer.py:206-329.  image (P,D,h,w); image_pe (P or 1,D,h,w); tokens (P,Nt,D).
    Returns (tokens, image) with image as (P, hw, D)."""
    keys = image.flatten(2).transpose(1, 2)
    kpe = image_pe.flatten(2).transpose(1, 2)
    tpe = tokens                                    # token "PE" is the initial token tensor (:240)
    qs = tokens
    for l in range(depth):
        lp = f"{pre}.layers.{l}"
        if l == 0:                                  # layer 0 REPLACES the tokens (:302-303)
            qs = dec_attention(w, lp + ".self_attn", qs, qs, qs, heads)
        else:
            qq = qs + tpe
            qs = qs + dec_attention(w, lp + ".self_attn", qq, qq, qs, heads)
        qs = layer_norm(w, lp + ".norm1", qs, 1e-5)
        qs = qs + dec_attention(w, lp + ".cross_attn_token_to_image", qs + tpe, keys + kpe, keys, heads)
        qs = layer_norm(w, lp + ".norm2", qs, 1e-5)
        m = linear(w, lp + ".mlp.lin2", torch.relu(linear(w, lp + ".mlp.lin1", qs)))
        qs = layer_norm(w, lp + ".norm3", qs + m, 1e-5)
        keys = keys + dec_attention(w, lp + ".cross_attn_image_to_token", keys + kpe, qs + tpe, qs, heads)
        keys = layer_norm(w, lp + ".norm4", keys, 1e-5)
    qs = qs + dec_attention(w, pre + ".final_attn_token_to_image", qs + tpe, keys + kpe, keys, heads)
    qs = layer_norm(w, pre + ".norm_final_attn", qs, 1e-5)
    return qs, keys


# --------------------------------------------------------------------------------------
# positional encoding  (models/prompt_encoder.py:187-233)
# --------------------------------------------------------------------------------------
def pe_encode(w: W, coords01: Tensor, pre: str = "prompt_encoder.pe_layer") -> Tensor:
    gm = w[pre + ".positional_encoding_gaussian_matrix"]
    c = (2.0 * coords01.to(gm.dtype) - 1.0) @ gm
    c = 2.0 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(w: W, g: int) -> Tensor:
    """(1, D, g, g): encodes pixel centres (i+0.5)/g, channel order [sin | cos]."""
    ctr = (torch.arange(g, dtype=torch.float32) + 0.5) / g
    yy, xx = torch.meshgrid(ctr, ctr, indexing="ij")
    return pe_encode(w, torch.stack([xx, yy], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def point_pe(w: W, xy: Tensor, image_size: int) -> Tensor:
    return pe_encode(w, xy / float(image_size))


# --------------------------------------------------------------------------------------
# prompt encoder  (models/prompt_encoder.py:396-827)
# --------------------------------------------------------------------------------------
def mask_downscale(w: W, masks: Tensor, pre: str = "prompt_encoder.mask_downscaling") -> Tensor:
    """(P,1,Hm,Wm) -> (P,D,Hm/4,Wm/4): conv2x2s2 -> LN2d -> GELU -> conv2x2s2 -> LN2d -> GELU -> conv1x1 (:61-69)."""
    x = F.conv2d(masks, w[pre + ".0.weight"], w[pre + ".0.bias"], stride=2)
    x = gelu(layer_norm_2d(w, pre + ".1", x))
    x = F.conv2d(x, w[pre + ".3.weight"], w[pre + ".3.bias"], stride=2)
    x = gelu(layer_norm_2d(w, pre + ".4", x))
    return F.conv2d(x, w[pre + ".6.weight"], w[pre + ".6.bias"])


def embed_sparse(w: W, geo: LamGeometry, b: int, m: int, c: int,
                 points: Optional[Tuple[Tensor, Tensor]], boxes: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """-> (B*M, C*Ns, D) before sparse_embedding_attention.  prompt_encoder.py:564-611, 83-114, 648-669."""
    pre = "prompt_encoder"
    d = geo.embed_dim
    s = geo.image_size
    parts: List[Tensor] = []
    if points is not None:
        xy, lab = points
        xy = xy.reshape(b * m * c, -1, 2) + 0.5
        lab = lab.reshape(b * m * c, -1)
        if boxes is None:  # one extra token at (0,0) with label -1 (== NEGATIVE in this code base)
            xy = torch.cat([xy, torch.zeros(xy.shape[0], 1, 2)], dim=1)
            lab = torch.cat([lab, -torch.ones(lab.shape[0], 1, dtype=lab.dtype)], dim=1)
        emb = point_pe(w, xy, s)
        null = lab == 0
        emb = torch.where(null[..., None], w[pre + ".not_a_point_embed.weight"].expand_as(emb), emb)
        emb = emb + (lab == -1)[..., None] * w[pre + ".point_embeddings.0.weight"]
        emb = emb + (lab == 1)[..., None] * w[pre + ".point_embeddings.1.weight"]
        parts.append(emb)
    if boxes is not None:
        bx, bf = boxes
        nb = bx.shape[3]
        corners = (bx.reshape(b * m * c * nb, 2, 2) + 0.5)
        emb = point_pe(w, corners, s)
        emb = torch.stack([emb[:, 0] + w[pre + ".point_embeddings.2.weight"][0],
                           emb[:, 1] + w[pre + ".point_embeddings.3.weight"][0]], dim=1)
        emb = emb.reshape(b * m * c, nb * 2, d)
        # the reference tiles the flags ([f0..fN-1, f0..fN-1]) against interleaved corner tokens (:661-667)
        flags2 = bf.reshape(b * m * c, nb).repeat(1, 2)
        emb = torch.where((flags2 == 0)[..., None], w[pre + ".not_a_point_embed.weight"].expand_as(emb), emb)
        parts.append(emb)
    if not parts:
        sp = w[pre + ".no_sparse_embedding.weight"].expand(b * m * c, 1, d)
    else:
        sp = torch.cat(parts, dim=1)
    ns = sp.shape[1]
    return sp.reshape(b * m, c * ns, d)


def prompt_encoder(w: W, geo: LamGeometry, support_emb: Tensor,
                   points: Optional[Tuple[Tensor, Tensor]], boxes: Optional[Tuple[Tensor, Tensor]],
                   masks: Optional[Tuple[Tensor, Tensor]], flag_examples: Tensor,
                   selected_rows: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """support_emb (B,M,D,g,g).  Returns class_embeddings (B,C,D), class_examples_embeddings (B,M,C,D),
    class_examples_src (P,D,g,g), flag_examples.  prompt_encoder.py:752-827."""
    pre = "prompt_encoder"
    d = geo.embed_dim
    heads = geo.dec_heads
    first = points[0] if points is not None else boxes[0] if boxes is not None else masks[0] if masks is not None else None
    if first is None:
        raise ValueError("No prompts provided")
    b, m, c = first.shape[:3]
    p = b * m * c
    g = support_emb.shape[-1]

    sparse = embed_sparse(w, geo, b, m, c, points, boxes)
    sparse = attention_mlp_block(w, pre + ".sparse_embedding_attention", sparse, heads)
    ns = sparse.shape[1] // c
    sparse = sparse.reshape(b, m, c, ns, d)

    if masks is not None:
        mk, mf = masks
        dense = mask_downscale(w, mk.reshape(p, 1, mk.shape[-2], mk.shape[-1]))
        missing = (mf.reshape(p) == 0).view(p, 1, 1, 1)
        dense = torch.where(missing, w[pre + ".not_a_mask_embed.weight"].view(1, d, 1, 1).expand_as(dense), dense)
    else:
        dense = w[pre + ".no_mask_embed.weight"].view(1, d, 1, 1).expand(p, d, g, g)
    if dense.shape[-2:] != support_emb.shape[-2:]:
        dense = F.interpolate(dense, size=support_emb.shape[-2:], mode="bilinear", align_corners=False)

    src = support_emb.unsqueeze(2).expand(b, m, c, d, g, g).reshape(p, d, g, g) + dense
    pos = dense_pe(w, g)

    if geo.class_encoder_bank:
        assert selected_rows is not None, "RandomMatrixEncoder is stochastic; pass selected_rows"
        ce = w[pre + ".class_encoder.pos_embedding"][0, 0, selected_rows]      # (C, D)
        sparse = sparse + ce.view(1, 1, c, 1, d)
        src = (src.view(b, m, c, d, g, g) + ce.view(1, 1, c, d, 1, 1)).reshape(p, d, g, g)

    _, keys = two_way_transformer(w, pre + ".transformer", src, pos, sparse.reshape(p, ns, d), heads)
    src_out = keys.transpose(1, 2).reshape(p, d, g, g)

    emb = keys.mean(dim=1).view(b, m, c, d)
    fe = flag_examples
    if geo.class_attention:
        emb = attention_mlp_block(w, pre + ".class_attention", emb.reshape(b * m, c, d), heads).view(b, m, c, d)
    if geo.example_attention:
        e2 = emb.permute(0, 2, 1, 3).reshape(b * c, m, d)
        emb = attention_mlp_block(w, pre + ".example_attention", e2, heads).view(b, c, m, d).permute(0, 2, 1, 3)
    if geo.example_class_attention:
        emb = attention_mlp_block(w, pre + ".class_example_attention", emb.reshape(b, m * c, d), heads).view(b, m, c, d)
    fe_f = fe.to(emb.dtype).unsqueeze(-1)
    denom = fe_f.sum(dim=1)
    denom = torch.where(denom == 0, torch.ones_like(denom), denom)
    cls = (emb * fe_f).sum(dim=1) / denom
    return {"flag_examples": fe, "class_embeddings": cls, "class_examples_embeddings": emb,
            "class_examples_src": src_out}


# --------------------------------------------------------------------------------------
# mask decoder  (models/mask_decoder.py:169-363, 776-804)
# --------------------------------------------------------------------------------------
def mask_decoder(w: W, geo: LamGeometry, query_emb: Tensor, class_emb: Tensor) -> Tensor:
    """query_emb (B,D,g,g), class_emb (B,C,D) -> logits (B,C,4g,4g)."""
    pre = "mask_decoder"
    b, d, g, _ = query_emb.shape
    toks, img = two_way_transformer(w, pre + ".transformer", query_emb, dense_pe(w, g), class_emb, geo.dec_heads)
    feat = img.transpose(1, 2).reshape(b, d, g, g)
    # class_mlp: 3 Linear layers, ReLU between
    pr = torch.relu(linear(w, pre + ".class_mlp.layers.0", toks))
    pr = torch.relu(linear(w, pre + ".class_mlp.layers.1", pr))
    pr = linear(w, pre + ".class_mlp.layers.2", pr)
    # output_upscaling: ConvT(k2,s2) -> LN2d -> GELU -> ConvT(k2,s2)  (no trailing activation)
    up = F.conv_transpose2d(feat, w[pre + ".output_upscaling.0.weight"], w[pre + ".output_upscaling.0.bias"], stride=2)
    up = gelu(layer_norm_2d(w, pre + ".output_upscaling.1", up))
    up = F.conv_transpose2d(up, w[pre + ".output_upscaling.3.weight"], w[pre + ".output_upscaling.3.bias"], stride=2)
    if geo.spatial_convs:
        for i in range(geo.spatial_convs):
            up = F.conv2d(up, w[f"{pre}.spatial_convs.{3 * i}.weight"], w[f"{pre}.spatial_convs.{3 * i}.bias"], padding=1)
            if i < geo.spatial_convs - 1:
                up = gelu(layer_norm_2d(w, f"{pre}.spatial_convs.{3 * i + 1}", up))
    bb, ch, hh, ww = up.shape
    return (pr @ up.view(bb, ch, hh * ww)).view(bb, -1, hh, ww)


# --------------------------------------------------------------------------------------
# post-processing  (models/lam.py:383-453, 92-93; data/utils.py:441-449)
# --------------------------------------------------------------------------------------
def preprocess_shape(h: int, wd: int, side: int) -> Tuple[int, int]:
    s = side * 1.0 / max(h, wd)
    return int(h * s + 0.5), int(wd * s + 0.5)


def postprocess(geo: LamGeometry, logits: Tensor, dims: Tensor, flag_gts: Optional[Tensor] = None) -> Tensor:
    """logits (B,C,4g,4g), dims (B,M+1,2) int (H,W) -> (B,C,Hmax,Wmax) with -inf padding (class 0 padded with 0)."""
    s = geo.image_size
    hmax, wmax = [int(v) for v in dims.reshape(-1, 2).max(dim=0).values.tolist()]
    qdims = [(int(h), int(wd)) for h, wd in dims[:, 0, :].tolist()]
    big = F.interpolate(logits, (s, s), mode="bilinear", align_corners=False)
    outs = []
    for i, (h, wd) in enumerate(qdims):
        one = big[i]
        if geo.custom_preprocess:
            ph, pw = preprocess_shape(h, wd, s)
            one = one[:, :ph, :pw]
        one = F.interpolate(one.unsqueeze(0), (h, wd), mode="bilinear", align_corners=False)
        outs.append(F.pad(one, (0, wmax - wd, 0, hmax - h), value=float("-inf")))
    out = torch.cat(outs)
    bg = out[:, 0]
    bg[bg == float("-inf")] = 0
    if flag_gts is not None:
        out[flag_gts.logical_not()] = float("-inf")
    return out


# --------------------------------------------------------------------------------------
# whole model  (models/lam.py:57-136, 349-381)
# --------------------------------------------------------------------------------------
def encode_images(w: W, geo: LamGeometry, images: Tensor) -> Tensor:
    if geo.encoder == "sam":
        emb = sam_encoder(w, geo, images)
    elif geo.encoder == "hf":
        emb = hf_vit_encoder(w, geo, images)
    else:
        raise ValueError("geometry has no image encoder")
    return emb


def episode_embeddings(w: W, geo: LamGeometry, batch: Dict[str, Tensor]) -> Tensor:
    """-> (B, N, D, g, g).  lam.py:138-170."""
    if "embeddings" in batch:
        e = batch["embeddings"]
        b, n = e.shape[:2]
        e = e.flatten(0, 1)
    elif "images" in batch:
        im = batch["images"]
        b, n = im.shape[:2]
        e = encode_images(w, geo, im.flatten(0, 1))
    else:
        raise ValueError("Either 'images' or 'embeddings' must be provided.")
    if geo.lam_neck:
        e = conv_neck(w, "neck", e)
    return e.view(b, n, *e.shape[1:])


def select_prompts(batch: Dict[str, Tensor]):
    """A prompt type is dropped entirely when all its flags are zero (lam.py:214-239)."""
    pts = bxs = msk = None
    if "prompt_points" in batch and bool((batch["flag_points"] != 0).any()):
        pts = (batch["prompt_points"], batch["flag_points"])
    if "prompt_bboxes" in batch and bool((batch["flag_bboxes"] != 0).any()):
        bxs = (batch["prompt_bboxes"], batch["flag_bboxes"])
    if "prompt_masks" in batch and bool((batch["flag_masks"] != 0).any()):
        msk = (batch["prompt_masks"], batch["flag_masks"])
    return pts, bxs, msk


def lam_forward(w: W, geo: LamGeometry, batch: Dict[str, Tensor],
                selected_rows: Optional[Tensor] = None, stages: Optional[dict] = None) -> Dict[str, Tensor]:
    emb = episode_embeddings(w, geo, batch)
    pts, bxs, msk = select_prompts(batch)
    pe = prompt_encoder(w, geo, emb[:, 1:], pts, bxs, msk, batch["flag_examples"], selected_rows)
    low = mask_decoder(w, geo, emb[:, 0], pe["class_embeddings"])
    out = postprocess(geo, low, batch["dims"], batch.get("flag_gts"))
    if stages is not None:
        stages.update(embeddings=emb, class_embeddings=pe["class_embeddings"],
                      class_examples_src=pe["class_examples_src"], low_res_logits=low)
    return {"logits": out, "class_examples_embeddings": pe["class_examples_embeddings"]}


def generate_class_embeddings(w: W, geo: LamGeometry, examples: Dict[str, Tensor],
                              selected_rows: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """lam.py:349-360: supports only; every image of the dict is a support."""
    if "embeddings" in examples:
        emb = examples["embeddings"]
        if geo.lam_neck:
            pass  # the reference does NOT apply the neck to precomputed embeddings on this path (lam.py:199-200)
    else:
        im = examples["images"]
        b, n = im.shape[:2]
        e = encode_images(w, geo, im.flatten(0, 1))
        if geo.lam_neck:
            e = conv_neck(w, "neck", e)
        emb = e.view(b, n, *e.shape[1:])
    pts, bxs, msk = select_prompts(examples)
    return prompt_encoder(w, geo, emb, pts, bxs, msk, examples["flag_examples"], selected_rows)


def predict(w: W, geo: LamGeometry, batch: Dict[str, Tensor], class_embeddings: Dict[str, Tensor]) -> Tensor:
    """lam.py:362-381: query-only encode + decode against cached prototypes."""
    if "embeddings" in batch:
        emb = batch["embeddings"]
    else:
        im = batch["images"]
        b, n = im.shape[:2]
        e = encode_images(w, geo, im.flatten(0, 1))
        if geo.lam_neck:
            e = conv_neck(w, "neck", e)
        emb = e.view(b, n, *e.shape[1:])
    low = mask_decoder(w, geo, emb[:, 0], class_embeddings["class_embeddings"])
    return postprocess(geo, low, batch["dims"].unsqueeze(1))
