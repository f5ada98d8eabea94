"""Shared parity cases: model geometry + seeded synthetic episode for each.

Used by tools/make_golden.py (reference -> fixtures), the CPU oracle tests and the GPU
parity tests, so all three see the same weights and inputs.
"""
from __future__ import annotations

from labelanything_amd.config import EncoderSpec, LamConfig, register_encoder
from oracle.lam_oracle import LamGeometry

# reduced-size encoders (test-only geometries; same code paths as the full ones)
register_encoder("sam_tiny", EncoderSpec("sam", dim=128, depth=2, heads=2, mlp=512, img_size=224,
                                         global_idx=(1,), window=8, out_chans=96))
register_encoder("hf_tiny", EncoderSpec("hf", dim=128, depth=2, heads=2, mlp=512, img_size=224))
# 8x8 patches (facebook/dino-vitb8 style) and a wide SAM stack (ViT-L style: 1024 wide, 16 heads, 14x14 windows)
register_encoder("hf_tiny_p8", EncoderSpec("hf", dim=128, depth=2, heads=2, mlp=512, patch=8, img_size=224))
# heads that are not 64 wide: SAM ViT-H style 80-wide heads (zero-padded to 128 by the packed weights) and 32-wide HF heads (-> 64)
register_encoder("sam_hd80_tiny", EncoderSpec("sam", dim=160, depth=2, heads=2, mlp=320, img_size=224, global_idx=(1,), window=8, out_chans=96))
register_encoder("hf_hd32_tiny", EncoderSpec("hf", dim=128, depth=2, heads=4, mlp=512, img_size=224))
register_encoder("sam_wide", EncoderSpec("sam", dim=1024, depth=2, heads=16, mlp=4096, img_size=448,
                                         global_idx=(1,), window=14, out_chans=256))


def geometry_for(cfg: LamConfig) -> LamGeometry:
    spec = cfg.encoder_spec
    kw = dict(
        encoder=None, image_size=cfg.image_size, patch=cfg.vit_patch_size,
        image_embed_dim=cfg.image_embed_dim, embed_dim=cfg.embed_dim,
        class_attention=cfg.class_attention, example_attention=cfg.example_attention,
        example_class_attention=cfg.example_class_attention, class_encoder_bank=cfg.bank_size,
        spatial_convs=cfg.spatial_convs, custom_preprocess=cfg.custom_preprocess,
        dec_heads=cfg.dec_heads, dec_mlp=cfg.dec_mlp, mask_in_chans=cfg.mask_in_chans,
    )
    if spec is not None:
        kw.update(encoder=spec.kind, enc_dim=spec.dim, enc_depth=spec.depth, enc_heads=spec.heads,
                  enc_mlp=spec.mlp, global_idx=tuple(spec.global_idx), window=spec.window,
                  sam_neck=cfg.use_vit_sam_neck, sam_out=spec.out_chans, hf_pos_grid=spec.pos_grid)
    return LamGeometry(**kw)


CASES = {
    # SAM-style encoder (padded 8x8 windows on a 14x14 grid + one global block), SAM neck 128->96,
    # LAM neck 96->64, all three prompt types, class encoder on, non-square original sizes.
    "sam_tiny_2w2s_all_prompts": dict(
        cfg=LamConfig(encoder="sam_tiny", image_size=224, image_embed_dim=96, embed_dim=64, spatial_convs=3,
                      class_encoder={"name": "RandomMatrixEncoder", "bank_size": 10, "embed_dim": 64},
                      custom_preprocess=True),
        weight_seed=11,
        episode=dict(batch=1, n_ways=2, k_shots=2, image_size=224, seed=101,
                     prompts=("mask", "point", "box"), dims=[[150, 200]] * 5),
    ),
    # HF plain ViT at 240 px (pos-emb bicubic 14->15), LAM neck 128->64, mask-only 1-way 1-shot (cfg1 shape).
    "hf_tiny_1w1s_masks": dict(
        cfg=LamConfig(encoder="hf_tiny", image_size=240, image_embed_dim=128, embed_dim=64, spatial_convs=3,
                      example_class_attention=False, custom_preprocess=False),
        weight_seed=12,
        episode=dict(batch=2, n_ways=1, k_shots=1, image_size=240, seed=102, prompts=("mask",)),
    ),
    # decoder-only with the real D=256 decoder, precomputed 256-ch embeddings (cfg4 shape, reduced grid),
    # class encoder on, one missing mask, points only on top.
    "novit_d256_2w3s": dict(
        cfg=LamConfig(encoder=None, use_vit=False, image_size=256, image_embed_dim=256, embed_dim=256,
                      spatial_convs=3, class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256},
                      custom_preprocess=True),
        weight_seed=13,
        episode=dict(batch=1, n_ways=2, k_shots=3, image_size=256, seed=103, prompts=("mask", "point"),
                     embeddings_channels=256, grid=16, drop_mask_of=(0, 1, 1), dims=[[256, 192]] * 7),
    ),
    # published SAM-1024 style decoder: D=512 fed by 768-ch pre-neck features through the LAM neck,
    # example_attention instead of example_class_attention (parameters/validation/old/COCO_Fold0_sam.yaml:255-270)
    "novit_d512_neck_1w2s": dict(
        cfg=LamConfig(encoder=None, use_vit=False, image_size=256, image_embed_dim=768, embed_dim=512,
                      spatial_convs=3, example_attention=True, example_class_attention=False,
                      class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 512}),
        weight_seed=14,
        episode=dict(batch=1, n_ways=1, k_shots=2, image_size=256, seed=104, prompts=("mask",),
                     embeddings_channels=768, grid=16),
    ),
    # BASELINE cfg2: SAM ViT-B 1024 + LAM decoder, 1-way 1-shot (the benchmark configuration). Slow on CPU.
    "cfg2_sam_b_1024_1w1s": dict(
        cfg=LamConfig(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3,
                      custom_preprocess=False),
        weight_seed=2,
        episode=dict(batch=1, n_ways=1, k_shots=1, image_size=1024, seed=1234, prompts=("mask",)),
        slow=True, store_full_logits=False, store_query_embedding=False, oracle_tol=1e-4,
    ),
    # BASELINE cfg1: ViT-MAE-B 480 1-way 1-shot (published upload geometry, push_to_hub.ipynb cell 2).
    "cfg1_mae_b_480_1w1s": dict(
        cfg=LamConfig(encoder="vit_b_mae", image_size=480, image_embed_dim=768, embed_dim=256, spatial_convs=3,
                      example_class_attention=False, custom_preprocess=False),
        weight_seed=1,
        episode=dict(batch=1, n_ways=1, k_shots=1, image_size=480, seed=1234, prompts=("mask",)),
        slow=True, store_full_logits=False, store_query_embedding=False, oracle_tol=1e-4,
    ),
}


# Training-step fixture (tests/golden/train_step.safetensors, tools/make_golden_train.py): decoder-only model with the LAM neck
# (96 -> 64), D = 64, class encoder on, masks + points + boxes, 2-way 2-shot, two episodes, non-square original sizes.
TRAIN_CASE = dict(
    cfg=LamConfig(encoder=None, use_vit=False, image_size=128, image_embed_dim=96, embed_dim=64, spatial_convs=3,
                  class_encoder={"name": "RandomMatrixEncoder", "bank_size": 10, "embed_dim": 64}, custom_preprocess=True),
    weight_seed=21,
    episode=dict(batch=2, n_ways=2, k_shots=2, image_size=128, seed=121, prompts=("mask", "point", "box"),
                 embeddings_channels=96, grid=8, dims=[[100, 128]] * 5),
    lr=1e-3, weight_decay=1e-2, steps=3, warmup=2,
)


# Same fixture with a TRAINABLE image encoder (tests/golden/train_step_encoder.safetensors): the reduced HF ViT at 240 px (position
# embeddings resampled 14 -> 15) + LAM neck + decoder, masks + points, no freeze_backbone - parameters/trainval/coco20i/mae_noembs.yaml.
TRAIN_ENC_CASE = dict(
    cfg=LamConfig(encoder="hf_tiny", image_size=240, image_embed_dim=128, embed_dim=64, spatial_convs=3, example_class_attention=False,
                  custom_preprocess=False),
    weight_seed=22,
    episode=dict(batch=1, n_ways=1, k_shots=2, image_size=240, seed=122, prompts=("mask", "point")),
    lr=1e-3, weight_decay=1e-2, steps=2, warmup=2,
)


# Training fixture with a TRAINABLE SAM ViTDet encoder (tests/golden/train_step_sam.safetensors): the reduced SAM stack of the forward
# fixtures (one padded 8 x 8 window block on the 14 x 14 grid + one global block, rel-pos tables, absolute position embedding), SAM neck
# 128 -> 96, LAM neck 96 -> 64, masks + points; nothing frozen - what ``lam_b`` trains when ``freeze_backbone`` is absent (lam.py:321-347).
TRAIN_SAM_CASE = dict(
    cfg=LamConfig(encoder="sam_tiny", image_size=224, image_embed_dim=96, embed_dim=64, spatial_convs=3, custom_preprocess=False),
    weight_seed=23,
    episode=dict(batch=1, n_ways=1, k_shots=2, image_size=224, seed=123, prompts=("mask", "point")),
    lr=1e-3, weight_decay=1e-2, steps=2, warmup=2,
)


# The same with SAM ViT-H style 80-wide heads (build_encoder.py:9-28: 1280 / 16; here 160 / 2) - tests/golden/train_step_sam_hd80.safetensors:
# forward AND backward run the heads zero-padded to 128 columns; ``lam_h`` with nothing frozen.
TRAIN_SAM_HD80_CASE = dict(
    cfg=LamConfig(encoder="sam_hd80_tiny", image_size=224, image_embed_dim=96, embed_dim=64, spatial_convs=3, custom_preprocess=False),
    weight_seed=24,
    episode=dict(batch=1, n_ways=1, k_shots=2, image_size=224, seed=124, prompts=("mask", "point")),
    lr=1e-3, weight_decay=1e-2, steps=2, warmup=2,
)
