"""Parameter inventory and seeded random initialisation for the hot-path model.

Key names and shapes are the reference's ``Lam.state_dict()`` layout (SURVEY.md
Appendix A; captured from /root/reference/label_anything/models/build_lam.py:96-235,
image_encoder.py, prompt_encoder.py, mask_decoder.py).  HF-ViT keys use the
transformers 4.x names the published checkpoints contain.

Pure CPU torch; used by the product (random-init models for bench / smoke), by
tests and by tools/make_golden.py (so the reference, the oracle and the HIP path
all see bit-identical weights).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .config import LamConfig, EncoderSpec

Shapes = Dict[str, Tuple[int, ...]]


def _attn_shapes(s: Shapes, pre: str, d: int, internal: int) -> None:
    for p in ("q_proj", "k_proj", "v_proj"):
        s[f"{pre}.{p}.weight"] = (internal, d)
        s[f"{pre}.{p}.bias"] = (internal,)
    s[f"{pre}.out_proj.weight"] = (d, internal)
    s[f"{pre}.out_proj.bias"] = (d,)


def _ln_shapes(s: Shapes, pre: str, d: int) -> None:
    s[f"{pre}.weight"] = (d,)
    s[f"{pre}.bias"] = (d,)


def _two_way_shapes(s: Shapes, pre: str, d: int, mlp: int, depth: int = 2) -> None:
    for l in range(depth):
        lp = f"{pre}.layers.{l}"
        _attn_shapes(s, lp + ".self_attn", d, d)
        _attn_shapes(s, lp + ".cross_attn_token_to_image", d, d // 2)
        _attn_shapes(s, lp + ".cross_attn_image_to_token", d, d // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            _ln_shapes(s, f"{lp}.{n}", d)
        s[lp + ".mlp.lin1.weight"] = (mlp, d)
        s[lp + ".mlp.lin1.bias"] = (mlp,)
        s[lp + ".mlp.lin2.weight"] = (d, mlp)
        s[lp + ".mlp.lin2.bias"] = (d,)
    _attn_shapes(s, pre + ".final_attn_token_to_image", d, d // 2)
    _ln_shapes(s, pre + ".norm_final_attn", d)


def _attn_mlp_block_shapes(s: Shapes, pre: str, d: int, internal: int, mlp: int) -> None:
    _ln_shapes(s, pre + ".norm", d)
    _attn_shapes(s, pre + ".attn", d, internal)
    s[pre + ".mlp.lin1.weight"] = (mlp, d)
    s[pre + ".mlp.lin1.bias"] = (mlp,)
    s[pre + ".mlp.lin2.weight"] = (d, mlp)
    s[pre + ".mlp.lin2.bias"] = (d,)


def _conv_neck_shapes(s: Shapes, pre: str, cin: int, cout: int) -> None:
    s[pre + ".0.weight"] = (cout, cin, 1, 1)
    _ln_shapes(s, pre + ".1", cout)
    s[pre + ".2.weight"] = (cout, cout, 3, 3)
    _ln_shapes(s, pre + ".3", cout)


def encoder_shapes(spec: EncoderSpec, sam_neck: bool = True, pre: str = "image_encoder") -> Shapes:
    s: Shapes = {}
    e = spec.dim
    if spec.kind == "sam":
        g = spec.img_size // spec.patch
        s[pre + ".pos_embed"] = (1, g, g, e)
        s[pre + ".patch_embed.proj.weight"] = (e, 3, spec.patch, spec.patch)
        s[pre + ".patch_embed.proj.bias"] = (e,)
        for i in range(spec.depth):
            bp = f"{pre}.blocks.{i}"
            _ln_shapes(s, bp + ".norm1", e)
            _ln_shapes(s, bp + ".norm2", e)
            s[bp + ".attn.qkv.weight"] = (3 * e, e)
            s[bp + ".attn.qkv.bias"] = (3 * e,)
            size = g if i in spec.global_idx else spec.window
            s[bp + ".attn.rel_pos_h"] = (2 * size - 1, spec.head_dim)
            s[bp + ".attn.rel_pos_w"] = (2 * size - 1, spec.head_dim)
            s[bp + ".attn.proj.weight"] = (e, e)
            s[bp + ".attn.proj.bias"] = (e,)
            s[bp + ".mlp.lin1.weight"] = (spec.mlp, e)
            s[bp + ".mlp.lin1.bias"] = (spec.mlp,)
            s[bp + ".mlp.lin2.weight"] = (e, spec.mlp)
            s[bp + ".mlp.lin2.bias"] = (e,)
        # the reference always instantiates the SAM neck, even when project_last_hidden=False
        _conv_neck_shapes(s, pre + ".neck", e, spec.out_chans)
    elif spec.kind == "hf":
        s[pre + ".embeddings.cls_token"] = (1, 1, e)
        s[pre + ".embeddings.position_embeddings"] = (1, 1 + spec.pos_grid ** 2, e)
        s[pre + ".embeddings.patch_embeddings.projection.weight"] = (e, 3, spec.patch, spec.patch)
        s[pre + ".embeddings.patch_embeddings.projection.bias"] = (e,)
        for i in range(spec.depth):
            lp = f"{pre}.encoder.layer.{i}"
            _ln_shapes(s, lp + ".layernorm_before", e)
            _ln_shapes(s, lp + ".layernorm_after", e)
            for n in ("query", "key", "value"):
                s[f"{lp}.attention.attention.{n}.weight"] = (e, e)
                s[f"{lp}.attention.attention.{n}.bias"] = (e,)
            s[lp + ".attention.output.dense.weight"] = (e, e)
            s[lp + ".attention.output.dense.bias"] = (e,)
            s[lp + ".intermediate.dense.weight"] = (spec.mlp, e)
            s[lp + ".intermediate.dense.bias"] = (spec.mlp,)
            s[lp + ".output.dense.weight"] = (e, spec.mlp)
            s[lp + ".output.dense.bias"] = (e,)
        _ln_shapes(s, pre + ".layernorm", e)
    else:
        raise ValueError(spec.kind)
    return s


def decoder_shapes(cfg: LamConfig) -> Shapes:
    """neck + prompt_encoder + mask_decoder (everything outside image_encoder.*)."""
    s: Shapes = {}
    d = cfg.embed_dim
    mlp = cfg.dec_mlp
    if cfg.lam_neck:
        _conv_neck_shapes(s, "neck", cfg.image_embed_dim, d)
    pe = "prompt_encoder"
    s[pe + ".pe_layer.positional_encoding_gaussian_matrix"] = (2, d // 2)
    for i in range(4):
        s[f"{pe}.point_embeddings.{i}.weight"] = (1, d)
    for n in ("not_a_point_embed", "no_mask_embed", "no_sparse_embedding", "not_a_mask_embed"):
        s[f"{pe}.{n}.weight"] = (1, d)
    c1, c2 = cfg.mask_in_chans // 4, cfg.mask_in_chans
    s[pe + ".mask_downscaling.0.weight"] = (c1, 1, 2, 2)
    s[pe + ".mask_downscaling.0.bias"] = (c1,)
    _ln_shapes(s, pe + ".mask_downscaling.1", c1)
    s[pe + ".mask_downscaling.3.weight"] = (c2, c1, 2, 2)
    s[pe + ".mask_downscaling.3.bias"] = (c2,)
    _ln_shapes(s, pe + ".mask_downscaling.4", c2)
    s[pe + ".mask_downscaling.6.weight"] = (d, c2, 1, 1)
    s[pe + ".mask_downscaling.6.bias"] = (d,)
    _two_way_shapes(s, pe + ".transformer", d, mlp)
    if cfg.bank_size:
        s[pe + ".class_encoder.pos_embedding"] = (1, 1, cfg.bank_size, d)
    _attn_mlp_block_shapes(s, pe + ".sparse_embedding_attention", d, d, mlp)
    if cfg.class_attention:
        _attn_mlp_block_shapes(s, pe + ".class_attention", d, d // 2, mlp)
    if cfg.example_class_attention:
        _attn_mlp_block_shapes(s, pe + ".class_example_attention", d, d // 2, mlp)
    if cfg.example_attention:
        _attn_mlp_block_shapes(s, pe + ".example_attention", d, d // 2, mlp)
    md = "mask_decoder"
    _two_way_shapes(s, md + ".transformer", d, mlp)
    s[md + ".output_upscaling.0.weight"] = (d, d // 4, 2, 2)      # ConvTranspose2d: (Cin, Cout, kh, kw)
    s[md + ".output_upscaling.0.bias"] = (d // 4,)
    _ln_shapes(s, md + ".output_upscaling.1", d // 4)
    s[md + ".output_upscaling.3.weight"] = (d // 4, d // 8, 2, 2)
    s[md + ".output_upscaling.3.bias"] = (d // 8,)
    s[md + ".class_mlp.layers.0.weight"] = (d, d)
    s[md + ".class_mlp.layers.0.bias"] = (d,)
    s[md + ".class_mlp.layers.1.weight"] = (d, d)
    s[md + ".class_mlp.layers.1.bias"] = (d,)
    s[md + ".class_mlp.layers.2.weight"] = (d // 8, d)
    s[md + ".class_mlp.layers.2.bias"] = (d // 8,)
    if cfg.spatial_convs:
        ch = d // 8
        for i in range(cfg.spatial_convs):
            s[f"{md}.spatial_convs.{3 * i}.weight"] = (ch, ch, 3, 3)
            s[f"{md}.spatial_convs.{3 * i}.bias"] = (ch,)
            if i < cfg.spatial_convs - 1:
                _ln_shapes(s, f"{md}.spatial_convs.{3 * i + 1}", ch)
    return s


def model_shapes(cfg: LamConfig) -> Shapes:
    s: Shapes = {}
    spec = cfg.encoder_spec
    if spec is not None:
        s.update(encoder_shapes(spec))
    s.update(decoder_shapes(cfg))
    return s


_NORM_TAGS = (".norm", ".layernorm", "neck.1.", "neck.3.", "mask_downscaling.1.", "mask_downscaling.4.",
              "output_upscaling.1.", "spatial_convs.1.", "spatial_convs.4.", "spatial_convs.7.")


def _is_norm(name: str) -> bool:
    return any(t in name for t in _NORM_TAGS)


def init_state_dict(cfg: LamConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded, fan-in scaled random weights.  Every tensor is non-trivial (rel-pos tables and
    pos_embed are NOT zero-initialised as in the reference, image_encoder.py:72-74,236-237, so
    positional bugs cannot hide)."""
    gen = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in model_shapes(cfg).items():
        if _is_norm(name):
            if name.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
            else:
                t = 0.05 * torch.randn(shape, generator=gen)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=gen)
        elif name.endswith("positional_encoding_gaussian_matrix"):
            t = torch.randn(shape, generator=gen)
        elif "rel_pos" in name:
            t = 0.1 * torch.randn(shape, generator=gen)
        elif name.endswith("pos_embed") or "position_embeddings" in name or "cls_token" in name \
                or "pos_embedding" in name:
            t = 0.02 * torch.randn(shape, generator=gen)
        elif len(shape) == 2 and shape[0] == 1:      # nn.Embedding(1, D) rows
            t = torch.randn(shape, generator=gen)
        else:
            if "output_upscaling" in name and len(shape) == 4:
                fan_in = shape[0]                    # ConvTranspose2d (Cin, Cout, k, k), stride == k: one tap per output
            else:
                fan_in = int(math.prod(shape[1:]))
            t = torch.randn(shape, generator=gen) / math.sqrt(max(fan_in, 1))
        out[name] = t.float().contiguous()
    return out
