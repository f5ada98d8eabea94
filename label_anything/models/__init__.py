"""Mirror of /root/reference/label_anything/models/__init__.py for the hot path: every LabelAnything entry of its ``model_registry``
(:33-60) plus the encoder-only entries (``**ENCODERS``) that ``preprocess.py:105-107`` indexes with ``model_registry[encoder_name]``.
The baseline few-shot models (dcama, fptrans, panet, ...), SAM itself and the multilevel / pyramid variant are out of scope
 and absent."""
from labelanything_amd.models import (  # noqa: F401
    ENCODERS, ImageEncoder, LabelAnything, Lam, build_encoder, build_lam, build_lam_dino_b8, build_lam_no_vit, build_lam_vit_b,
    build_lam_vit_b_imagenet_i21k, build_lam_vit_h, build_lam_vit_l, build_lam_vit_mae_b, build_vit_b, build_vit_b_imagenet_i21k,
    build_vit_b_mae, build_vit_dino_b8, build_vit_h, build_vit_l,
)

# registry name -> builder; the names are the contract (experiment configs and the embeddings CLI index the dict by them)
model_registry = dict(
    lam=build_lam, lam_no_vit=build_lam_no_vit,
    lam_h=build_lam_vit_h, lam_l=build_lam_vit_l, lam_b=build_lam_vit_b,
    lam_mae_b=build_lam_vit_mae_b, lam_dino_b8=build_lam_dino_b8, lam_b_imagenet_i21k=build_lam_vit_b_imagenet_i21k,
)
model_registry.update(ENCODERS)        # encoder-only entries: vit_h, vit_l, vit_b, vit_b_mae, vit_dino_b8
