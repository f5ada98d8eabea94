#!/usr/bin/env python
"""Golden vectors for the TRAINING step: the REFERENCE itself is imported, its ``WrapperModule`` (model +
``LabelAnythingLoss({'focal': {'weight': 1.0}}, class_weighting=True)``, experiment/utils.py:266-303) is run on a seeded episode,
``loss.backward()`` and ``steps`` iterations of torch.optim.AdamW + HF ``constant_with_warmup`` (experiment/utils.py:53-100,
mae_noembs.yaml:24-37) are applied to ``Lam.get_learnable_params({'freeze_backbone': True})``.  Stored in
tests/golden/train_step.safetensors: the loss of every step, per-tensor L2 norms of the first gradient and of the total parameter
change for EVERY learnable tensor, and the full first gradient + final value of a handful of tensors.  The oracle's autograd
(oracle/lam_oracle.py + oracle/loss_oracle.py) is checked against the reference's gradient before anything is written.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_train.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.make_golden as MG          # noqa: E402  (installs the stub finder, puts /root/reference first)

import torch                            # noqa: E402
from safetensors.torch import save_file  # noqa: E402

FULL = ["neck.0.weight", "neck.3.bias", "prompt_encoder.mask_downscaling.0.weight", "prompt_encoder.mask_downscaling.4.weight",
        "prompt_encoder.point_embeddings.1.weight", "prompt_encoder.not_a_mask_embed.weight", "prompt_encoder.class_encoder.pos_embedding",
        "prompt_encoder.transformer.layers.1.cross_attn_image_to_token.k_proj.weight", "prompt_encoder.transformer.layers.0.mlp.lin1.bias",
        "prompt_encoder.class_example_attention.attn.out_proj.weight", "prompt_encoder.sparse_embedding_attention.norm.weight",
        "mask_decoder.transformer.final_attn_token_to_image.q_proj.weight", "mask_decoder.transformer.layers.1.norm4.bias",
        "mask_decoder.output_upscaling.0.weight", "mask_decoder.output_upscaling.3.bias", "mask_decoder.spatial_convs.3.weight",
        "mask_decoder.class_mlp.layers.2.weight"]


FULL_ENC = ["image_encoder.embeddings.patch_embeddings.projection.weight", "image_encoder.embeddings.position_embeddings",
            "image_encoder.embeddings.cls_token", "image_encoder.encoder.layer.0.attention.attention.query.weight",
            "image_encoder.encoder.layer.0.layernorm_before.weight", "image_encoder.encoder.layer.1.attention.attention.value.weight",
            "image_encoder.encoder.layer.1.attention.output.dense.bias", "image_encoder.encoder.layer.1.intermediate.dense.weight",
            "image_encoder.encoder.layer.1.output.dense.weight", "image_encoder.layernorm.weight", "neck.0.weight",
            "prompt_encoder.transformer.layers.0.cross_attn_token_to_image.v_proj.weight", "mask_decoder.output_upscaling.0.weight"]


FULL_SAM = ["image_encoder.pos_embed", "image_encoder.patch_embed.proj.weight", "image_encoder.blocks.0.attn.rel_pos_h",
            "image_encoder.blocks.0.attn.rel_pos_w", "image_encoder.blocks.0.attn.qkv.weight", "image_encoder.blocks.0.attn.qkv.bias",
            "image_encoder.blocks.0.norm1.weight", "image_encoder.blocks.1.attn.rel_pos_h", "image_encoder.blocks.1.attn.rel_pos_w",
            "image_encoder.blocks.1.attn.proj.weight", "image_encoder.blocks.1.mlp.lin1.weight", "image_encoder.blocks.1.mlp.lin2.bias",
            "image_encoder.neck.0.weight", "image_encoder.neck.2.weight", "image_encoder.neck.3.bias", "neck.0.weight",
            "mask_decoder.output_upscaling.0.weight"]


def main():
    from tests.cases import TRAIN_CASE, TRAIN_ENC_CASE, TRAIN_SAM_CASE, TRAIN_SAM_HD80_CASE
    which = sys.argv[1:] or ["decoder", "hf", "sam", "sam_hd80"]
    if "decoder" in which:
        run(TRAIN_CASE, "train_step", FULL, seed_gt=9)
    if "hf" in which:
        run(TRAIN_ENC_CASE, "train_step_encoder", FULL_ENC, seed_gt=11)
    if "sam" in which:          # the SAM ViTDet stack trainable (window + global rel-pos attention, position embedding, SAM neck)
        run(TRAIN_SAM_CASE, "train_step_sam", FULL_SAM, seed_gt=13)
    if "sam_hd80" in which:     # ... with 80-wide heads (SAM ViT-H style): the padded-head backward against the reference
        run(TRAIN_SAM_HD80_CASE, "train_step_sam_hd80", FULL_SAM, seed_gt=14)


def _to_4x(k: str) -> str:
    """This container's transformers names the ViT tensors differently from the 4.x names the repo (and the published checkpoints)
    use; inverse of tools/make_golden.py:_hf_4x_to_local."""
    if not k.startswith("image_encoder."):
        return k
    k = k.replace("image_encoder.encoder.layers.", "image_encoder.layers.")
    if "image_encoder.layers." not in k:
        return k
    return (k.replace("image_encoder.layers.", "image_encoder.encoder.layer.")
             .replace("attention.q_proj", "attention.attention.query").replace("attention.k_proj", "attention.attention.key")
             .replace("attention.v_proj", "attention.attention.value").replace("attention.o_proj", "attention.output.dense")
             .replace("mlp.fc1", "intermediate.dense").replace("mlp.fc2", "output.dense"))


def run(case, out_name, full, seed_gt):
    from label_anything.experiment.utils import WrapperModule
    from label_anything.loss import LabelAnythingLoss
    from transformers import get_scheduler
    from tests.cases import geometry_for
    from tests.test_train_gpu import make_gt
    from oracle import lam_oracle as O
    from oracle import loss_oracle as LO
    from labelanything_amd.episodes import make_episode

    lam, sd = MG.build_reference(case)
    lam.train()                                     # dropout is 0 everywhere on this path; train() as the reference's loop does
    cfg = case["cfg"]
    batch = make_episode(**case["episode"])
    c = batch["flag_examples"].shape[2]
    gt = make_gt(batch, c, seed=seed_gt)
    gr = torch.Generator().manual_seed(case["weight_seed"] + 7)
    rows = None
    if cfg.bank_size:
        rows = torch.cat([torch.zeros(1, dtype=torch.long), torch.randperm(cfg.bank_size - 1, generator=gr)[: c - 1] + 1])
        lam.prompt_encoder.class_encoder.sample_rows = lambda C, device, _r=rows: _r.to(device)
    model = WrapperModule(lam, LabelAnythingLoss({"focal": {"weight": 1.0}}, class_weighting=True))
    # no freeze_backbone (mae_noembs.yaml): every parameter is learnable, the image encoder included (lam.py:321-347)
    params = model.get_learnable_params({})
    named = {_to_4x(k): p for k, p in lam.named_parameters()}
    assert len(params) == len(named) and all(k in sd for k in named), [k for k in named if k not in sd][:4]
    opt = torch.optim.AdamW(params, lr=case["lr"], weight_decay=case["weight_decay"])
    sched = get_scheduler("constant_with_warmup", opt, num_warmup_steps=case["warmup"], num_training_steps=100)
    out = {"gt": gt}
    if rows is not None:
        out["selected_rows"] = rows
    start = {k: p.detach().clone() for k, p in named.items()}
    losses = []
    for step in range(case["steps"]):
        res = model(batch, gt)
        loss = res["loss"]["value"]
        loss.backward()
        losses.append(float(loss))
        if step == 0:
            grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in named.items()}
            # the oracle's autograd must reproduce the reference's gradient before it may serve as the on-box checker
            w = {k: v.clone().requires_grad_(v.is_floating_point() and "gaussian" not in k) for k, v in sd.items()}
            o = O.lam_forward(w, geometry_for(cfg), batch, selected_rows=rows)
            ol, _ = LO.focal_objective(o["logits"], gt)
            ol.backward()
            assert abs(float(ol) - float(loss)) <= 1e-6 * max(1.0, abs(float(loss))), (float(ol), float(loss))
            gmax = max(float(g.abs().max()) for g in grads.values())
            for k, g in grads.items():
                og = w[k].grad if w[k].grad is not None else torch.zeros_like(g)
                err = float((og - g).abs().max()) / max(float(g.abs().max()), 1e-2 * gmax)
                assert err <= 2e-4, (k, err)
            out["logits0"] = res["logits"].detach().clone()
        opt.step()
        sched.step()
        opt.zero_grad()
    keys = sorted(named)
    out["loss"] = torch.tensor(losses)
    out["grad_norm"] = torch.stack([grads[k].norm() for k in keys])
    out["delta_norm"] = torch.stack([(named[k].detach() - start[k]).norm() for k in keys])
    for k in full:
        out["grad." + k] = grads[k].contiguous()
        out["final." + k] = named[k].detach().clone().contiguous()
    save_file(out, os.path.join(ROOT, "tests", "golden", out_name + ".safetensors"))
    with open(os.path.join(ROOT, "tests", "golden", out_name + ".json"), "w") as fh:
        import json
        json.dump({"keys": keys, "losses": losses, "generated_by": "tools/make_golden_train.py", "torch": torch.__version__}, fh, indent=1)
    print("losses", losses, "tensors", len(keys), "bytes", sum(v.numel() * v.element_size() for v in out.values()))


if __name__ == "__main__":
    main()
