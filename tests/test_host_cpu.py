"""CPU: host logic, C-ABI surface and the model's parameter / config I/O (no kernels are launched)."""
import ctypes
import os
import re
import tempfile

import pytest
import torch

from labelanything_amd import _lib
from labelanything_amd.config import LamConfig, config_from_kwargs
from labelanything_amd.episodes import make_episode
from labelanything_amd.weights import init_state_dict, model_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "la_hip.h")).read()
    declared = set(re.findall(r"\b(la_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libla_hip.so does not export {name}"
    assert declared == set(_lib.EXPORTS) | {"la_last_error", "la_version"}
    assert lib.la_version() >= 1


def test_state_dict_layout_matches_reference_inventory():
    """LabelAnything(encoder='vit_b', D=256, class encoder on) = 411 tensors, 99.03 M params."""
    cfg = LamConfig(encoder="vit_b", spatial_convs=3, class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256})
    shapes = model_shapes(cfg)
    n_params = sum(int(torch.Size(s).numel()) for s in shapes.values())
    assert len(shapes) == 411
    assert abs(n_params / 1e6 - 99.03) < 0.02
    assert shapes["image_encoder.blocks.2.attn.rel_pos_h"] == (127, 64)
    assert shapes["image_encoder.blocks.0.attn.rel_pos_h"] == (27, 64)
    assert shapes["mask_decoder.output_upscaling.0.weight"] == (256, 64, 2, 2)
    # decoder-only MAE-480 geometry: 10.14 M trainable params
    cfg2 = LamConfig(encoder=None, use_vit=False, image_embed_dim=768, embed_dim=256, image_size=480, spatial_convs=3,
                     class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256})
    n2 = sum(int(torch.Size(s).numel()) for s in model_shapes(cfg2).values())
    assert abs(n2 / 1e6 - 10.14) < 0.02


def test_label_anything_config_roundtrip_and_hub_format():
    from label_anything.models import LabelAnything
    m = LabelAnything(encoder=None, use_vit=False, image_size=256, embed_dim=64, image_embed_dim=96, spatial_convs=3)
    assert m.config["embed_dim"] == 64 and m.config["encoder"] is None
    assert all(k.startswith("model.") for k in m.state_dict())
    with tempfile.TemporaryDirectory() as d:
        m.save_pretrained(d)
        assert {"config.json", "model.safetensors"} <= set(os.listdir(d))
        m2 = LabelAnything.from_pretrained(d)
    a, b = m.state_dict(), m2.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_off_path_switches_are_rejected():
    with pytest.raises(NotImplementedError):
        config_from_kwargs(encoder="vit_b", few_type="Affinity")
    with pytest.raises(NotImplementedError):
        config_from_kwargs(encoder="vit_b", fusion_transformer="OneWayTransformer")
    with pytest.raises(TypeError):
        config_from_kwargs(encoder="vit_b", not_an_argument=1)


def test_dropout_is_accepted_for_inference_and_refused_by_the_trainer():
    """models/common.py:25-32,68-75: the reference applies nn.Dropout(p) in training mode only.  Inference accepts the switch (eval-mode
    no-op, like the reference); the trainer has no dropout node and must say so instead of silently training a different model."""
    from labelanything_amd.models import Lam
    from labelanything_amd.train import LamTrainer
    cfg = config_from_kwargs(encoder=None, use_vit=False, image_size=64, embed_dim=64, image_embed_dim=64, spatial_convs=3, dropout=0.1)
    assert cfg.dropout == pytest.approx(0.1)
    assert config_from_kwargs(encoder=None, use_vit=False, image_size=64, embed_dim=64, image_embed_dim=64).dropout == 0.0
    with pytest.raises(NotImplementedError, match="dropout"):
        LamTrainer(Lam(cfg))


def test_model_refuses_to_run_without_gpu_or_library(monkeypatch):
    from labelanything_amd.models import Lam
    lam = Lam(LamConfig(encoder=None, use_vit=False, image_size=64, embed_dim=64, image_embed_dim=64, spatial_convs=3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lam({"embeddings": torch.zeros(1, 2, 64, 4, 4)})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libla_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="HIP extension .* is missing"):
        _lib.lib()


def test_synthetic_episode_schema():
    b = make_episode(batch=2, n_ways=2, k_shots=2, image_size=64, seed=3, prompts=("mask", "point", "box"))
    assert b["images"].shape == (2, 5, 3, 64, 64)
    assert b["prompt_masks"].shape == (2, 4, 3, 256, 256) and b["flag_masks"].shape == (2, 4, 3)
    assert b["prompt_points"].shape[-1] == 2 and b["prompt_bboxes"].shape[-1] == 4
    assert bool((b["flag_examples"][:, :, 0] == 1).all())           # background always present
    assert b["dims"].shape == (2, 5, 2) and b["flag_gts"].shape == (2, 3)
    b2 = make_episode(batch=2, n_ways=2, k_shots=2, image_size=64, seed=3, prompts=("mask", "point", "box"))
    assert all(torch.equal(b[k], b2[k]) for k in b)


def test_hf_key_renaming_from_transformers5():
    from labelanything_amd.models import _hf5_to_hf4
    sd = {"image_encoder.layers.3.attention.q_proj.weight": torch.zeros(1),
          "image_encoder.layers.3.mlp.fc2.bias": torch.zeros(1), "image_encoder.pooler.dense.weight": torch.zeros(1)}
    out = _hf5_to_hf4(sd)
    assert set(out) == {"image_encoder.encoder.layer.3.attention.attention.query.weight",
                        "image_encoder.encoder.layer.3.output.dense.bias"}


def test_bench_workloads_describe_valid_models_and_episodes():
    """bench.py --workload: every BASELINE geometry builds a LamConfig and a synthetic episode of the right shape (no GPU)."""
    import bench
    from labelanything_amd.config import LamConfig
    from labelanything_amd.episodes import make_episode
    assert set(bench.WORKLOADS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg3_train", "cfg2_train"}
    for name, w in bench.WORKLOADS.items():
        cfg = LamConfig(**w["model"])
        ep = dict(w["episode"])
        small = make_episode(batch=1, seed=1, prompts=("mask",), **{**ep, "image_size": 64, **({"grid": 4} if "grid" in ep else {})})
        m = ep["n_ways"] * ep["k_shots"]
        key = "embeddings" if "embeddings_channels" in ep else "images"
        assert small[key].shape[:2] == (1, m + 1) and small["prompt_masks"].shape[:3] == (1, m, ep["n_ways"] + 1)
        assert (cfg.encoder_spec is None) == (name == "cfg4")


def test_sam_checkpoint_initialisation_and_encoder_less_checkpoints(tmp_path):
    """use_sam_checkpoint (models/lam.py:241-319): encoder + the SAM-shaped decoder parts come from the checkpoint, SAM's
    mask-decoder transformer seeds BOTH two-way transformers, everything else keeps its own initialisation; mismatches raise.
    ignore_encoder_checkpoint (utils/utils.py:111-139): only image_encoder.* keys may be missing."""
    import torch
    from safetensors.torch import save_file
    from labelanything_amd.config import EncoderSpec, register_encoder
    from labelanything_amd.models import build_lam
    register_encoder("sam_micro", EncoderSpec("sam", dim=64, depth=1, heads=1, mlp=128, img_size=64, global_idx=(0,), window=0, out_chans=256))
    kw = dict(encoder="sam_micro", image_size=64, image_embed_dim=256, embed_dim=256, spatial_convs=3)
    donor = build_lam(seed=5, **kw)
    sam = {k: v.clone() for k, v in donor.state_dict().items()
           if k.startswith(("image_encoder.", "prompt_encoder.pe_layer.", "prompt_encoder.point_embeddings.", "prompt_encoder.not_a_point_embed.",
                            "prompt_encoder.mask_downscaling.", "prompt_encoder.no_mask_embed.", "mask_decoder.transformer.",
                            "mask_decoder.output_upscaling."))}
    path = str(tmp_path / "sam_like.safetensors")
    save_file({k: v.contiguous() for k, v in sam.items()}, path)
    fresh = build_lam(seed=6, **kw).state_dict()
    lam = build_lam(seed=6, checkpoint=path, use_sam_checkpoint=True, **kw)
    got = lam.state_dict()
    for k in ("image_encoder.blocks.0.attn.qkv.weight", "prompt_encoder.mask_downscaling.0.weight", "mask_decoder.output_upscaling.3.bias"):
        assert torch.equal(got[k], sam[k])
    t = "transformer.layers.1.cross_attn_image_to_token.k_proj.weight"
    assert torch.equal(got["prompt_encoder." + t], sam["mask_decoder." + t])          # SAM's decoder transformer seeds both
    assert torch.equal(got["mask_decoder." + t], sam["mask_decoder." + t])
    for k in ("mask_decoder.class_mlp.layers.0.weight", "prompt_encoder.class_example_attention.attn.q_proj.weight", "mask_decoder.spatial_convs.0.weight"):
        assert torch.equal(got[k], fresh[k])                                            # not part of a SAM checkpoint
    # a checkpoint that lacks part of a seeded module is an error, as in the reference's strict per-module loads
    bad = {k: v for k, v in sam.items() if k != "mask_decoder.output_upscaling.3.bias"}
    badp = str(tmp_path / "bad.safetensors")
    save_file({k: v.contiguous() for k, v in bad.items()}, badp)
    with pytest.raises(RuntimeError, match="output_upscaling"):
        build_lam(seed=6, checkpoint=badp, use_sam_checkpoint=True, **kw)
    # ignore_encoder_checkpoint: decoder-only checkpoint accepted, anything else missing is an error
    dec = {k: v.contiguous() for k, v in donor.state_dict().items() if not k.startswith("image_encoder.")}
    decp = str(tmp_path / "decoder_only.safetensors")
    save_file(dec, decp)
    lam2 = build_lam(seed=7, checkpoint=decp, ignore_encoder_checkpoint=True, **kw)
    assert torch.equal(lam2.state_dict()["mask_decoder.class_mlp.layers.0.weight"], dec["mask_decoder.class_mlp.layers.0.weight"])
    with pytest.raises(RuntimeError):
        build_lam(seed=7, checkpoint=decp, **kw)                                        # strict by default
    dec.pop("mask_decoder.class_mlp.layers.0.weight")
    save_file(dec, decp)
    with pytest.raises(RuntimeError, match="Missing keys"):
        build_lam(seed=7, checkpoint=decp, ignore_encoder_checkpoint=True, **kw)


def test_has_config_records_arguments():
    from labelanything_amd.models import LabelAnything
    m = LabelAnything(encoder=None, use_vit=False, image_size=256, embed_dim=64, image_embed_dim=64, spatial_convs=3)
    assert m.config["embed_dim"] == 64 and m.config["use_vit"] is False and m.config["class_attention"] is False
    m2 = LabelAnything(config=dict(m.config))
    assert m2.config == m.config


def test_model_registry_has_every_on_path_reference_name(tmp_path):
    """label_anything/models/__init__.py:33-60 of the reference: the LabelAnything entries and the encoder-only ``**ENCODERS`` entries
    (``model_registry[encoder_name](checkpoint=..., use_sam_checkpoint=...)``, preprocess.py:105-107)."""
    import label_anything.models as M
    from safetensors.torch import save_file
    for name in ("lam", "lam_no_vit", "lam_h", "lam_l", "lam_b", "lam_mae_b", "lam_dino_b8", "lam_b_imagenet_i21k",
                 "vit_h", "vit_l", "vit_b", "vit_b_mae", "vit_dino_b8"):
        assert callable(M.model_registry[name]), name
    for fn in ("build_lam_vit_h", "build_lam_dino_b8", "build_lam_vit_b_imagenet_i21k", "build_vit_b", "build_vit_h", "build_vit_l",
               "build_vit_b_mae", "build_vit_dino_b8", "build_vit_b_imagenet_i21k", "build_encoder", "ENCODERS"):
        assert hasattr(M, fn), fn
    # geometry of the builders that were missing (shapes only: meta-free but cheap - the dino / i21k encoders are ViT-B sized)
    from labelanything_amd.config import ENCODER_SPECS
    assert ENCODER_SPECS["vit_b_imagenet_i21k"].kind == "hf" and ENCODER_SPECS["vit_b_imagenet_i21k"].patch == 16
    assert ENCODER_SPECS["vit_dino_b8"].patch == 8 and ENCODER_SPECS["vit_h"].dim == 1280
    # encoder-only builder: SAM-checkpoint key handling and strict loading, on a reduced geometry
    from tests.cases import CASES  # noqa: F401  (registers sam_tiny)
    enc = M.build_encoder("sam_tiny", seed=3)
    sd = enc.state_dict()
    assert "patch_embed.proj.weight" in sd and not any(k.startswith("image_encoder.") for k in sd)
    ck = str(tmp_path / "sam.safetensors")
    donor = {("image_encoder." + k): (v + 1).contiguous() for k, v in sd.items()}
    donor["mask_decoder.foo"] = torch.zeros(1)
    save_file(donor, ck)
    enc2 = M.build_encoder("sam_tiny", checkpoint=ck, use_sam_checkpoint=True)
    assert torch.equal(enc2.state_dict()["pos_embed"], sd["pos_embed"] + 1)
    with pytest.raises(RuntimeError):
        M.build_encoder("sam_tiny", checkpoint=ck)                   # prefixed keys without use_sam_checkpoint: strict load fails


def test_folded_layernorm_algebra_is_exact():
    """The identity behind LamEngine.norm_fold (image_encoder.py:181-197: Linear(LayerNorm(x))), in float64 on the CPU: with
    W' = W diag(gamma), c = row sums of W', b' = b + W beta and the row statistics (mean, rstd) of x,
        rstd (x W'^T - mean c) + b'  ==  LayerNorm(x) W^T + b
    - what the consumer GEMM's epilogue evaluates from the un-normalised operand - and the partial-sum form of the statistics
    (sum x, sum x^2 per 64-column slot, as the producer epilogue leaves them) reproduces mean and rstd."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    m, k, n, eps = 37, 768, 96, 1e-6
    x = (torch.randn(m, k, generator=g, dtype=torch.float64) * (0.5 + torch.rand(m, 1, generator=g, dtype=torch.float64) * 4)
         + torch.randn(m, 1, generator=g, dtype=torch.float64))
    gamma = 1 + 0.1 * torch.randn(k, generator=g, dtype=torch.float64)
    beta = 0.05 * torch.randn(k, generator=g, dtype=torch.float64)
    w = torch.randn(n, k, generator=g, dtype=torch.float64) / k ** 0.5
    b = 0.02 * torch.randn(n, generator=g, dtype=torch.float64)
    ref = F.linear(F.layer_norm(x, (k,), gamma, beta, eps), w, b)
    part = torch.stack([x.view(m, k // 64, 64).sum(2), (x * x).view(m, k // 64, 64).sum(2)], dim=2)
    mean = part[..., 0].sum(1, keepdim=True) / k
    rstd = (part[..., 1].sum(1, keepdim=True) / k - mean * mean + eps).rsqrt()
    wf = w * gamma
    out = rstd * (x @ wf.t() - mean * wf.sum(1)) + (b + w @ beta)
    assert float((out - ref).abs().max()) < 1e-11
    # the token-mean correction of a single-plane weight in the folded form multiplies the normalised rows BEFORE gamma
    z = (x - mean) * rstd
    lo = wf - wf.half().double()
    assert float((F.linear(z, lo) - (ref - (rstd * (x @ wf.half().double().t() - mean * wf.half().double().sum(1)) + (b + w @ beta)))).abs().max()) < 1e-11
