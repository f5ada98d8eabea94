"""Model geometry for the LabelAnything hot path.

Mirrors the reference's constructor surface: ``LabelAnything(**kwargs)``
(/root/reference/label_anything/models/build_lam.py:467-508) and the encoder
registry ``ENCODERS`` (models/build_encoder.py:9-28,144-152).  Pure Python; no
device code.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, Optional, Tuple


@dataclass(frozen=True)
class EncoderSpec:
    """One image-encoder geometry.

    kind "sam": ViTDet backbone with 14x14 windowed + global rel-pos attention
    (models/image_encoder.py).  kind "hf": plain pre-LN ViT with CLS token as in
    transformers.ViTModel (models/build_encoder.py:83-100).
    """

    kind: str
    dim: int
    depth: int
    heads: int
    mlp: int
    patch: int = 16
    img_size: int = 1024            # SAM: fixed input side (pos_embed size); HF: pretraining side (pos grid)
    global_idx: Tuple[int, ...] = ()
    window: int = 14
    out_chans: int = 256            # SAM neck output channels

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def pos_grid(self) -> int:
        return self.img_size // self.patch


ENCODER_SPECS: Dict[str, EncoderSpec] = {
    # models/build_encoder.py:9-28 (SAM ViT-B/L/H), _build_vit :43-80
    "vit_b": EncoderSpec("sam", 768, 12, 12, 3072, global_idx=(2, 5, 8, 11)),
    "vit_l": EncoderSpec("sam", 1024, 24, 16, 4096, global_idx=(5, 11, 17, 23)),
    "vit_h": EncoderSpec("sam", 1280, 32, 16, 5120, global_idx=(7, 15, 23, 31)),
    # facebook/vit-mae-base / -large geometry (README.md:147-165, parameters/trainval/coco/mael.yaml:49)
    "vit_b_mae": EncoderSpec("hf", 768, 12, 12, 3072, img_size=224),
    "vit_l_mae": EncoderSpec("hf", 1024, 24, 16, 4096, img_size=224),
    # facebook/dino-vitb8 (models/build_encoder.py:115-117): HF ViT-B with 8x8 patches (use vit_patch_size=8)
    "vit_dino_b8": EncoderSpec("hf", 768, 12, 12, 3072, patch=8, img_size=224),
    # google/vit-base-patch16-224-in21k (models/build_encoder.py:108-112): the plain HF ViT-B geometry
    "vit_b_imagenet_i21k": EncoderSpec("hf", 768, 12, 12, 3072, img_size=224),
}


def register_encoder(name: str, spec: EncoderSpec) -> None:
    """Register an extra geometry (used by tests for reduced-size encoders)."""
    ENCODER_SPECS[name] = spec


@dataclass
class LamConfig:
    """Keyword surface of ``LabelAnything.__init__`` (build_lam.py:470-498), on-path subset.

    Off-path switches (few_type=Affinity, OneWay/Identity fusion, binary, pyramids,
    segment_example_logits, conv_classification, class_embedding_dim, TokenPool) are
    accepted only at their default value; anything else raises NotImplementedError.
    """

    encoder: Optional[str] = "vit_b"
    use_vit: bool = True
    use_vit_sam_neck: bool = True
    image_embed_dim: int = 256
    embed_dim: int = 256
    image_size: int = 1024
    vit_patch_size: int = 16
    class_attention: bool = False
    example_attention: bool = False
    example_class_attention: bool = True
    spatial_convs: Optional[int] = None
    class_encoder: Optional[dict] = None      # {"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": D}
    custom_preprocess: bool = True
    # Dropout probability of the decoder-side MLP / attention blocks (models/common.py:25-32,68-75, build_lam.py:128).  An inference
    # no-op (eval mode); remembered so that LamTrainer can refuse to train a configuration whose reference applies it.
    dropout: float = 0.0
    # fixed in the reference for this path
    dec_heads: int = 8
    dec_mlp: int = 2048
    mask_in_chans: int = 16

    @property
    def grid(self) -> int:
        return self.image_size // self.vit_patch_size

    @property
    def lam_neck(self) -> bool:
        return self.image_embed_dim != self.embed_dim

    @property
    def encoder_spec(self) -> Optional[EncoderSpec]:
        if not self.use_vit or self.encoder is None:
            return None
        return ENCODER_SPECS[self.encoder]

    @property
    def bank_size(self) -> int:
        return int(self.class_encoder["bank_size"]) if self.class_encoder else 0


_OFF_PATH_DEFAULTS = dict(
    class_embedding_dim=None,
    encoder_attention_downsample_rate=2, decoder_attention_downsample_rate=2,
    classification_layer_downsample_rate=8, use_support_features_in_prompt_encoder=True,
    fusion_transformer="TwoWayTransformer", few_type="Prototype", class_fusion="sum",
    transformer_keys_are_images=True, transformer_feature_size=None,
    segment_example_logits=False, dropout=0.0, binary=False,
)


def config_from_kwargs(**kw) -> LamConfig:
    """Build a LamConfig from reference-style kwargs, rejecting off-path ablation switches."""
    kw = dict(kw)
    for k, dflt in _OFF_PATH_DEFAULTS.items():
        if k in kw:
            v = kw.pop(k)
            if k == "dropout":          # an inference no-op (eval mode), accepted like the reference does; LamTrainer raises on != 0
                kw["dropout"] = float(v or 0.0)
                continue
            if v != dflt:
                raise NotImplementedError(f"{k}={v!r} is an off-path ablation of the reference; only {dflt!r} is built")
    fields = LamConfig.__dataclass_fields__
    unknown = [k for k in kw if k not in fields]
    if unknown:
        raise TypeError(f"unexpected LabelAnything arguments: {unknown}")
    cfg = LamConfig(**kw)
    if cfg.class_encoder is not None and cfg.class_encoder.get("name") != "RandomMatrixEncoder":
        raise NotImplementedError("only RandomMatrixEncoder is built as class_encoder")
    return cfg
