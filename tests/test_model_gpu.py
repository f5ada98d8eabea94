"""GPU: the HIP path (through the C ABI) against the golden fixtures captured from the reference.

north_star tolerance: <= 1e-3 relative (max|a-b| / max|b|) on the logits with fp16/bf16 MFMA operands, argmax bit-exact.
The parity configuration is the DEFAULT one - fp16 operands, split-precision weight planes for the qkv / proj (and, for
narrow encoders, lin2) GEMMs, exact-fp32 patch embedding, necks and decoder (engine.resolve_precise) - and it
is held to 1e-3 on EVERY stage below (measured 4.7e-4 .. 7.0e-4 on the worst stage of the six cases, profiles/r02_parity.log).

Argmax: the fused argmax must equal torch.argmax of the logits the same launch wrote, bit for bit, and it must equal the
REFERENCE's argmax at every pixel whose reference top-2 margin exceeds 2 x the logit tolerance (two logits that each move
by <= tol can only swap when they were closer than 2 tol).  Random-weight models put 0.02-0.3 % of the pixels inside that
band; their number is printed and held, per case, to 2 x the count measured on MI355X (profiles/r04_parity.log).  Context:
"bit-exact" against the reference is unreachable for ANY re-ordered arithmetic on these random-weight fixtures - the fp32 CPU
oracle itself differs from the reference on 27 of the 1 048 576 pixels of cfg2 (tests/golden/cfg2_sam_b_1024_1w1s.json,
"argmax_ties") - so the assertion is: zero flips outside the band, and no more flips inside it than the measured error explains.

bf16 operands (8 mantissa bits) are a supported switch, NOT the parity configuration: their activation roundings alone
cost 3-6e-3 on the logits, so they are held to their own measured bound (x1.5) and documented as such.
"""
import pytest
import torch

from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from tests.cases import CASES
from tests.helpers import argmax_disagreement, load_golden, pct_rel_err, reference_logits, rel_err

pytestmark = pytest.mark.gpu

TOL = {
    torch.float16: dict(emb=1e-3, cls=1e-3, low=1e-3, logits=1e-3),
    torch.bfloat16: dict(emb=6.5e-3, cls=1.5e-3, low=9e-3, logits=8e-3),
}
# share of the pixels measured to flip inside the near-tie band, per case (profiles/r06_parity.log, default numerics - round 6: LayerNorm
# folded into the GEMMs for the two wide encoders, which draws another realisation of the weight roundings: cfg2 1.29e-3 -> 3.81e-3,
# cfg1 2.66e-3 -> 6.5e-4).  The TOLERANCE is the assertion above: no pixel whose reference top-2 margin exceeds the band may flip.  The
# count below is only a regression pin (2 x measured; results are deterministic, box to box the last bits are identical), and at least
# ARGMAX_FLOOR pixels for the decoder-only cases that measure zero
ARGMAX_MEASURED = {
    torch.float16: {"sam_tiny_2w2s_all_prompts": 8.7e-4, "hf_tiny_1w1s_masks": 2.1e-4, "novit_d256_2w3s": 0.0, "novit_d512_neck_1w2s": 0.0,
                    "cfg2_sam_b_1024_1w1s": 3.81e-3, "cfg1_mae_b_480_1w1s": 6.6e-4},
    torch.bfloat16: {"sam_tiny_2w2s_all_prompts": 4.6e-3, "hf_tiny_1w1s_masks": 2.5e-3, "novit_d256_2w3s": 0.0, "novit_d512_neck_1w2s": 0.0,
                     "cfg2_sam_b_1024_1w1s": 1.23e-2, "cfg1_mae_b_480_1w1s": 6.2e-3},
}
ARGMAX_FLOOR = 4
# 99.9th percentile of |a - b| / |b| over the logits with |b| > 1 % of max|b| (tests/helpers.pct_rel_err)
PCT_TOL = {torch.float16: 4e-2, torch.bfloat16: 3e-1}      # 2 x the worst measured (1.9e-2 cfg2, 1.5e-1 sam_tiny bf16: profiles/r05_parity.log)
ARGMAX_MARGIN = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}     # 2 x the logit tolerance


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", list(CASES))
def test_episode_matches_reference_fixture(name, dt):
    case = CASES[name]
    gold, _ = load_golden(name)
    lam = Lam(case["cfg"], seed=case["weight_seed"], compute_dtype=dt).cuda()
    lam.selected_rows = gold.get("selected_rows")
    batch = make_episode(**case["episode"])
    seg, pe = lam._forward(batch)
    out = lam.forward_argmax(batch)
    torch.cuda.synchronize()
    tol = TOL[dt]
    d = lam.cfg.embed_dim
    e32, b, n, g = lam._embeddings_nhwc(batch, True)
    q = e32.view(b, n, g * g, d)[:, 0].permute(0, 2, 1).reshape(b, d, g, g)
    if "query_embedding" in gold:
        assert rel_err(q, gold["query_embedding"]) <= tol["emb"]
    else:
        assert rel_err(q[:, ::8, ::4, ::4], gold["query_embedding_sample"]) <= tol["emb"]
    assert rel_err(pe["class_embeddings"], gold["class_embeddings"]) <= tol["cls"]
    assert rel_err(pe["class_examples_embeddings"], gold["class_examples_embeddings"]) <= tol["cls"]
    assert rel_err(seg, gold["low_res_logits"]) <= tol["low"]
    ref_logits = reference_logits(case, gold, batch)            # stored, or oracle post-processing of the stored low-res logits
    assert rel_err(out["logits"], ref_logits) <= tol["logits"]
    # second figure: 99.9th percentile of the element-wise relative error over the logits above 1 % of the maximum
    p999 = pct_rel_err(out["logits"], ref_logits)
    print(f"[p99.9 {name} {dt}] element-wise relative error of the logits with |ref| > 1 % of max: {p999:.3e} (bound {PCT_TOL[dt]:.1e})")
    assert p999 <= PCT_TOL[dt]
    am = out["argmax"].cpu()
    # the index kernel itself is exact: fused argmax == argmax of the logits it wrote
    assert torch.equal(out["logits"].argmax(dim=1).cpu(), am)
    # and equals the reference's argmax wherever the reference's top-2 margin is outside the tolerance band
    n_diff, n_real = argmax_disagreement(out["logits"], gold["argmax"].long(), ref_logits, margin_rel=ARGMAX_MARGIN[dt])
    assert n_real == 0, f"{n_real} of {n_diff} differing pixels have a reference margin above the tolerance band"
    allowed = max(ARGMAX_FLOOR, int(2 * ARGMAX_MEASURED[dt][name] * am.numel()))
    print(f"[argmax {name} {dt}] {n_diff} of {am.numel()} pixels flip inside the near-tie band (allowed {allowed}), {n_real} outside it")
    assert n_diff <= allowed, f"{n_diff} of {am.numel()} pixels flip inside the near-tie band (allowed {allowed} = 2 x measured)"
    assert out["logits"].shape == (b, gold["class_embeddings"].shape[1], *gold["argmax"].shape[-2:])


def test_plain_16bit_operands_are_faster_but_coarser():
    """precise=() (every encoder GEMM with single 16-bit planes) stays within 2e-3 - the round-1 configuration, kept as a
    switch; the split-precision default must be the tighter of the two."""
    name = "sam_tiny_2w2s_all_prompts"
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    errs = []
    for precise in ((), None):
        kw = {} if precise is None else {"precise": precise}
        lam = Lam(case["cfg"], seed=case["weight_seed"], **kw).cuda()
        lam.selected_rows = gold.get("selected_rows")
        errs.append(rel_err(lam(batch)["logits"], gold["logits"]))
    assert errs[0] <= 2.5e-3 and errs[1] <= 1e-3 and errs[1] < errs[0]
    with pytest.raises(ValueError, match="unknown precise groups"):
        Lam(case["cfg"], seed=1, precise=("mlp",)).cuda().engine()


def test_decoder_on_fp16_plane_pairs_keeps_fp32_level_parity():
    """decoder_dtype="f16x2": the image-side GEMM operands of the prompt encoder / mask decoder are fp16 plane pairs [hi | lo] against
    [W_hi | W_hi | W_lo] weights (three fast-MFMA products instead of the exact-fp32 MFMA): the decoder-only fixture (no encoder error
    at all) must stay at fp32 level, far below what one 16-bit plane gives."""
    name = "novit_d256_2w3s"
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    errs = {}
    for dd in ("f16x2", torch.float32, None):
        lam = Lam(case["cfg"], seed=case["weight_seed"], decoder_dtype=dd).cuda()
        lam.selected_rows = gold.get("selected_rows")
        errs[dd] = rel_err(lam(batch)["logits"], gold["logits"])
    assert errs["f16x2"] <= 2e-5 and errs[torch.float32] <= 2e-5
    assert errs[None] > 20 * errs["f16x2"]                   # the single 16-bit plane is the coarse one


def test_predict_with_cached_class_embeddings_matches_forward():
    """generate_class_embeddings + predict == forward for the same episode (lam.py:349-381)."""
    case = CASES["novit_d256_2w3s"]
    gold, _ = load_golden("novit_d256_2w3s")
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.selected_rows = gold["selected_rows"]
    batch = make_episode(**case["episode"])
    full = lam(batch)["logits"]
    examples = {k: (v[:, 1:] if k in ("embeddings", "dims") else v) for k, v in batch.items()}
    ce = lam.generate_class_embeddings(examples)
    assert ce["class_examples_src"].shape == (6 * 3, 256, 16, 16)
    q = {"embeddings": batch["embeddings"][:, :1], "dims": batch["dims"][:, 0]}
    pred = lam.predict(q, ce)
    torch.cuda.synchronize()
    assert rel_err(pred, full) < 1e-5


def test_missing_inputs_and_prompts_raise():
    case = CASES["novit_d256_2w3s"]
    lam = Lam(case["cfg"], seed=1).cuda()
    with pytest.raises(ValueError, match="Either 'images' or 'embeddings'"):
        lam({"dims": torch.zeros(1, 1, 2)})
    batch = make_episode(**case["episode"])
    for k in ("prompt_masks", "flag_masks", "prompt_points", "flag_points"):
        batch.pop(k)
    with pytest.raises(ValueError, match="No prompts provided"):
        lam(batch)


def test_image_encoder_handle_returns_nchw():
    case = CASES["sam_tiny_2w2s_all_prompts"]
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    out = lam.image_encoder(x)
    assert out.shape == (2, 96, 14, 14)
    both = lam.image_encoder(x, return_last_block_state=True)
    assert both["last_block_state"].shape == (2, 128, 14, 14)
    assert rel_err(both["last_hidden_state"], out) < 1e-6


def test_hip_graph_replay_matches_eager_and_tracks_new_inputs():
    """use_graphs: the captured launch sequence reproduces the eager result bit for bit, also after the inputs change."""
    case = CASES["sam_tiny_2w2s_all_prompts"]
    gold, _ = load_golden("sam_tiny_2w2s_all_prompts")
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.selected_rows = gold["selected_rows"]
    b1 = make_episode(**case["episode"])
    b2 = make_episode(**{**case["episode"], "seed": 777})
    e1, e2 = lam.forward_argmax(b1), lam.forward_argmax(b2)
    lam.use_graphs = True
    g1 = lam.forward_argmax(b1)       # capture
    g2 = lam.forward_argmax(b2)       # replay with new inputs
    g1b = lam.forward_argmax(b1)      # replay again
    torch.cuda.synchronize()
    for k in ("logits", "argmax", "class_examples_embeddings"):
        assert torch.equal(e1[k], g1[k]) and torch.equal(e2[k], g2[k]) and torch.equal(e1[k], g1b[k])
    assert len(lam._graphs) == 1


@pytest.mark.parametrize("prompts", [("point",), ("box",), ("mask", "box")])
def test_prompt_type_combinations_match_oracle(prompts):
    """Prompt types the golden cases do not cover (points only incl. the padding point, boxes only, masks + boxes),
    HIP path vs the CPU oracle on the same seeded episode (decoder-only D=256 geometry, exact-fp32 decoder => 1e-4)."""
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    case = CASES["novit_d256_2w3s"]
    cfg = case["cfg"]
    ep = dict(case["episode"])
    ep.update(prompts=prompts, seed=321)
    ep.pop("drop_mask_of", None)
    batch = make_episode(**ep)
    rows = torch.tensor([0, 17, 42])
    lam = Lam(cfg, seed=3).cuda()
    lam.selected_rows = rows
    out = lam(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(cfg, 3), geometry_for(cfg), batch, selected_rows=rows)
    assert rel_err(out["logits"], ref["logits"]) < 1e-4
    assert rel_err(out["class_examples_embeddings"], ref["class_examples_embeddings"]) < 1e-4


def test_class_and_example_attention_variants_match_oracle():
    """class_attention + example_attention switches (build_lam.py:192-194) on a ragged-size batch of 2 episodes."""
    from labelanything_amd.config import LamConfig
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    cfg = LamConfig(encoder=None, use_vit=False, image_size=128, image_embed_dim=64, embed_dim=64, spatial_convs=3,
                    class_attention=True, example_attention=True, example_class_attention=False, custom_preprocess=True)
    batch = make_episode(batch=2, n_ways=2, k_shots=2, image_size=128, seed=9, prompts=("mask", "point"), embeddings_channels=64,
                         grid=8, dims=[[100, 128], [128, 64], [90, 90], [128, 128], [77, 50]])
    batch["flag_gts"][1, 2] = False
    lam = Lam(cfg, seed=4).cuda()
    out = lam(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(cfg, 4), geometry_for(cfg), batch)
    assert out["logits"].shape == ref["logits"].shape == (2, 3, 128, 128)
    assert rel_err(out["logits"], ref["logits"]) < 1e-4


def test_many_pairs_5way_5shot_matches_oracle():
    """BASELINE cfg3 episode shape (5-way 5-shot: 26 images, 150 (support, class) pairs) on the reduced HF encoder:
    exercises the large-token paths (M > 32 token GEMMs on the fp32 MFMA kernel, 150-token class_example_attention)."""
    from labelanything_amd.config import LamConfig
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    cfg = LamConfig(encoder="hf_tiny", image_size=240, image_embed_dim=128, embed_dim=64, spatial_convs=3,
                    class_encoder={"name": "RandomMatrixEncoder", "bank_size": 20, "embed_dim": 64}, custom_preprocess=False)
    batch = make_episode(batch=1, n_ways=5, k_shots=5, image_size=240, seed=55, prompts=("mask", "point"))
    rows = torch.tensor([0, 3, 9, 11, 14, 18])
    lam = Lam(cfg, seed=8).cuda()
    lam.selected_rows = rows
    out = lam(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(cfg, 8), geometry_for(cfg), batch, selected_rows=rows)
    assert out["logits"].shape == (1, 6, 240, 240)
    e_cls, e_log = rel_err(out["class_examples_embeddings"], ref["class_examples_embeddings"]), rel_err(out["logits"], ref["logits"])
    print(f"many-pair episode: class_examples_embeddings {e_cls:.3e}, logits {e_log:.3e} (north_star bound 1e-3)")
    assert e_cls <= 1e-3 and e_log <= 1e-3, (e_cls, e_log)


def _encoder_vs_oracle(cfg, bn, seed):
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(bn, 3, cfg.image_size, cfg.image_size, generator=g)
    lam = Lam(cfg, seed=seed).cuda()
    got = lam.image_encoder(images.cuda()).float().cpu()
    sd, geo = init_state_dict(cfg, seed), geometry_for(cfg)
    with torch.no_grad():
        ref = O.sam_encoder(sd, geo, images) if cfg.encoder_spec.kind == "sam" else O.hf_vit_encoder(sd, geo, images)
    assert got.shape == ref.shape
    e = rel_err(got, ref)
    print(f"encoder {cfg.encoder} vs oracle: {e:.3e} (north_star bound 1e-3)")
    return e


def test_patch8_hf_encoder_matches_oracle():
    """facebook/dino-vitb8 geometry class (build_encoder.py:115-117): 8x8 patches, pos grid 28 resampled to 20."""
    from labelanything_amd.config import LamConfig
    cfg = LamConfig(encoder="hf_tiny_p8", image_size=160, vit_patch_size=8, image_embed_dim=128, embed_dim=64, spatial_convs=3,
                    custom_preprocess=False)
    assert _encoder_vs_oracle(cfg, 3, 21) <= 1e-3


def test_wide_sam_encoder_matches_oracle():
    """SAM ViT-L width (1024 channels, 16 heads of 64, 14x14 windows on a 28x28 grid, padded) through the same kernels."""
    from labelanything_amd.config import LamConfig
    cfg = LamConfig(encoder="sam_wide", image_size=448, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False)
    assert _encoder_vs_oracle(cfg, 2, 22) <= 1e-3


def test_sam_vit_h_style_80_wide_heads_match_oracle():
    """SAM ViT-H has 80-wide heads (1280 / 16, image_encoder.py:200-255).  The host zero-pads them to 128 for the attention
    kernels; windows (14x14, padded grid), a global block with la_relpos_terms (28x28 grid) and, at 1024 px, the in-kernel
    G = 64 path all have to reproduce the unpadded maths."""
    from labelanything_amd.config import EncoderSpec, LamConfig, register_encoder
    register_encoder("sam_hd80", EncoderSpec("sam", dim=160, depth=2, heads=2, mlp=320, img_size=448, global_idx=(1,), window=14, out_chans=64))
    cfg = LamConfig(encoder="sam_hd80", image_size=448, image_embed_dim=64, embed_dim=64, spatial_convs=3, custom_preprocess=False)
    assert _encoder_vs_oracle(cfg, 2, 23) <= 1e-3
    register_encoder("sam_hd80_1k", EncoderSpec("sam", dim=160, depth=2, heads=2, mlp=320, img_size=1024, global_idx=(1,), window=14, out_chans=64))
    cfg = LamConfig(encoder="sam_hd80_1k", image_size=1024, image_embed_dim=64, embed_dim=64, spatial_convs=3, custom_preprocess=False)
    assert _encoder_vs_oracle(cfg, 1, 24) <= 1e-3


def test_hf_encoder_with_32_wide_heads_is_padded_to_64():
    """Heads narrower than 64 (e.g. a 4-head 128-wide ViT) ride the same zero-padding."""
    from labelanything_amd.config import EncoderSpec, LamConfig, register_encoder
    register_encoder("hf_hd32", EncoderSpec("hf", dim=128, depth=2, heads=4, mlp=256, img_size=224))
    cfg = LamConfig(encoder="hf_hd32", image_size=160, image_embed_dim=128, embed_dim=64, spatial_convs=3, custom_preprocess=False)
    assert _encoder_vs_oracle(cfg, 2, 25) <= 1e-3


def test_cfg5_episode_shape_10way_5shot_matches_oracle():
    """BASELINE cfg5 episode shape (10-way 5-shot: 51 images, 550 (support, class) pairs, 11 classes) on the reduced HF encoder:
    the largest token counts of the path (550-token class_example_attention, 11-class decoder, mask + box + point prompts)."""
    from labelanything_amd.config import LamConfig
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    cfg = LamConfig(encoder="hf_tiny", image_size=160, image_embed_dim=128, embed_dim=64, spatial_convs=3,
                    class_encoder={"name": "RandomMatrixEncoder", "bank_size": 30, "embed_dim": 64}, custom_preprocess=False)
    batch = make_episode(batch=1, n_ways=10, k_shots=5, image_size=160, seed=77, prompts=("mask", "point", "box"))
    rows = torch.tensor([0, 2, 5, 7, 11, 13, 17, 19, 23, 26, 29])
    lam = Lam(cfg, seed=9).cuda()
    lam.selected_rows = rows
    out = lam(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(cfg, 9), geometry_for(cfg), batch, selected_rows=rows)
    assert out["logits"].shape == (1, 11, 160, 160)
    assert out["class_examples_embeddings"].shape == (1, 50, 11, 64)
    e_cls, e_log = rel_err(out["class_examples_embeddings"], ref["class_examples_embeddings"]), rel_err(out["logits"], ref["logits"])
    print(f"many-pair episode: class_examples_embeddings {e_cls:.3e}, logits {e_log:.3e} (north_star bound 1e-3)")
    assert e_cls <= 1e-3 and e_log <= 1e-3, (e_cls, e_log)


# ---- every BASELINE config at its own size ----------------------------------------------------------------------------------------
def _bench_model(workload, episodes=1, seed=7, **kw):
    import bench
    from labelanything_amd.config import LamConfig
    w = bench.WORKLOADS[workload]
    lam = Lam(LamConfig(**w["model"]), seed=3, **kw).cuda()
    batch = make_episode(batch=episodes, seed=seed, prompts=("mask",), **w["episode"])
    return lam, batch


def test_cfg4_full_size_matches_the_oracle():
    """BASELINE cfg4 at its own size: precomputed 256 x 64 x 64 embeddings, 2-way 5-shot (30 prompt pairs, hw = 4096), whole
    decoder + post-processing against the CPU oracle (44.6 GMAC: a few seconds of host time)."""
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    lam, batch = _bench_model("cfg4")
    out = lam.forward_argmax(batch)
    with torch.no_grad():
        ref = O.lam_forward(init_state_dict(lam.cfg, 3), geometry_for(lam.cfg), batch)
    assert out["logits"].shape == (1, 3, 1024, 1024)
    assert rel_err(out["logits"], ref["logits"]) <= 2e-5
    assert rel_err(out["class_examples_embeddings"], ref["class_examples_embeddings"]) <= 2e-5
    n_diff, n_real = argmax_disagreement(out["logits"], ref["logits"].argmax(1), ref["logits"], margin_rel=1e-4)
    assert n_real == 0 and n_diff <= 1e-3 * out["logits"][:, 0].numel()       # near-ties (margin < 1e-4 of the logit range) may swap


@pytest.mark.parametrize("workload", ["cfg3", "cfg5"])
def test_full_geometry_episode_properties(workload):
    """BASELINE cfg3 (MAE-B 480, 5-way 5-shot: 26 images, 150 pairs) and cfg5 (MAE-L 480, 10-way 5-shot: 51 images, 550 pairs) at
    full geometry.  The oracle needs minutes per episode here, so: shapes, finiteness, HIP-graph replay == eager launches bit for bit,
    and episode-shard invariance (two episodes in one batch == the same episodes one at a time: what N-GPU sharding relies on)."""
    lam, batch = _bench_model(workload, episodes=2)
    lam.selected_rows = torch.arange(batch["flag_examples"].shape[2])
    both = lam(batch)["logits"]
    c = batch["flag_examples"].shape[2]
    assert both.shape == (2, c, 480, 480) and bool(torch.isfinite(both).all())
    singles = [lam({k: v[i:i + 1] for k, v in batch.items()})["logits"] for i in range(2)]
    assert rel_err(both, torch.cat(singles)) <= 1e-5          # batched GEMM tiles see other row neighbours: accumulation-level only
    lam.use_graphs = True
    g1 = lam(batch)["logits"].clone()
    g2 = lam(batch)["logits"]
    assert torch.equal(g1, g2) and torch.equal(g1, both)


def test_cfg3_decoder_at_full_size_matches_the_oracle_on_device_embeddings():
    """cfg3 at full size, decoder side against the oracle: the HIP encoder's post-neck embeddings are handed to the CPU oracle's
    prompt encoder + mask decoder + post-processing (150 pairs x 900 positions: seconds), which must reproduce the HIP logits."""
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    lam, batch = _bench_model("cfg3")
    rows = torch.arange(batch["flag_examples"].shape[2])
    lam.selected_rows = rows
    out = lam(batch)
    e32, b, n, g = lam._embeddings_nhwc(batch, True)
    d = lam.cfg.embed_dim
    emb = e32.view(b, n, g * g, d).permute(0, 1, 3, 2).reshape(b, n, d, g, g).cpu()
    w = init_state_dict(lam.cfg, 3)
    geo = geometry_for(lam.cfg)
    with torch.no_grad():
        pts, bxs, msk = O.select_prompts(batch)
        pe = O.prompt_encoder(w, geo, emb[:, 1:], pts, bxs, msk, batch["flag_examples"], rows)
        low = O.mask_decoder(w, geo, emb[:, 0], pe["class_embeddings"])
        ref = O.postprocess(geo, low, batch["dims"], batch.get("flag_gts"))
    assert rel_err(out["class_examples_embeddings"], pe["class_examples_embeddings"]) <= 2e-5
    assert rel_err(out["logits"], ref) <= 5e-5


def test_cfg5_at_full_size_against_the_oracle_in_two_halves():
    """BASELINE cfg5 at its own size (ViT-MAE-L 480, 10-way 5-shot: 51 images, 550 prompt pairs) against the CPU oracle without the
    minutes a whole-episode oracle run would take: (a) the ViT-L encoder + LAM neck on THREE of the 51 images (the query and two
    supports; 313 GMAC each) against the oracle's embeddings, <= 1e-3; (b) the whole decoder side (550 pairs x 900 positions, 11 classes)
    by handing the HIP encoder's post-neck embeddings to the oracle's prompt encoder + mask decoder + post-processing, which must
    reproduce the HIP logits to fp32 accumulation accuracy."""
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    lam, batch = _bench_model("cfg5")
    rows = torch.arange(batch["flag_examples"].shape[2])
    lam.selected_rows = rows
    out = lam(batch)
    e32, b, n, g = lam._embeddings_nhwc(batch, True)
    d = lam.cfg.embed_dim
    emb = e32.view(b, n, g * g, d).permute(0, 1, 3, 2).reshape(b, n, d, g, g).cpu()
    w = init_state_dict(lam.cfg, 3)
    geo = geometry_for(lam.cfg)
    assert (b, n, g) == (1, 51, 30)
    with torch.no_grad():
        pick = [0, 1, 50]
        sub = {"images": batch["images"][:, pick].float().cpu()}
        ref_emb = O.episode_embeddings(w, geo, sub)[0]                                      # encoder + LAM neck: (3, D, g, g)
        e_err = rel_err(emb[0, pick], ref_emb)
        pts, bxs, msk = O.select_prompts(batch)
        pe = O.prompt_encoder(w, geo, emb[:, 1:], pts, bxs, msk, batch["flag_examples"], rows)
        low = O.mask_decoder(w, geo, emb[:, 0], pe["class_embeddings"])
        ref = O.postprocess(geo, low, batch["dims"], batch.get("flag_gts"))
    print(f"[cfg5 full size] embeddings of 3 images vs oracle {e_err:.3e}; decoder on device embeddings: logits {rel_err(out['logits'], ref):.3e}")
    assert e_err <= 1e-3
    assert rel_err(out["class_examples_embeddings"], pe["class_examples_embeddings"]) <= 2e-5
    assert rel_err(out["logits"], ref) <= 5e-5


# ---- BASELINE configs[4] as worded: "fp8 MFMA attention" ------------------------------------------------------------------------------
FP8_BOUND = 3.5e-3      # stated bound of the opt-in switch (measured 3.1e-3 on the cfg1 fixture, 2.7e-3 on a cfg5 episode: profiles/r03_attn_fp8.log)


def test_attn_fp8_on_the_cfg1_reference_fixture_holds_its_stated_bound():
    """``Lam.attn_fp8 = True`` (QK^T on v_mfma_scale_f32_32x32x64_f8f6f4 from an e4m3 copy of q | k) against the REFERENCE's logits
    of the cfg1 fixture: 3 mantissa bits on q and k cost 3e-3 - outside north_star's 1e-3, which is why the switch is opt-in - and must
    stay inside the bound the switch is documented with; the 16-bit path of the same model stays <= 1e-3; toggling the switch on a
    live model (graphs captured) takes effect and toggling it back restores the 16-bit result bit for bit."""
    name = "cfg1_mae_b_480_1w1s"
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    ref = reference_logits(case, gold, batch)
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.use_graphs = True
    base = lam(batch)["logits"].clone()
    lam.attn_fp8 = True
    out8 = lam(batch)["logits"].clone()
    lam.attn_fp8 = False
    again = lam(batch)["logits"]
    e16, e8 = rel_err(base, ref), rel_err(out8, ref)
    n_diff, n_real = argmax_disagreement(out8, gold["argmax"].long(), ref, margin_rel=2 * FP8_BOUND)
    print(f"[attn_fp8 cfg1] logits vs reference: 16-bit {e16:.3e}, fp8 QK^T {e8:.3e} (bound {FP8_BOUND}); argmax flips {n_diff}, outside the band {n_real}")
    assert e16 <= 1e-3
    assert e16 < e8 <= FP8_BOUND, (e16, e8)
    assert n_real == 0
    assert torch.equal(again, base)


def test_attn_fp8_on_a_cfg5_episode_at_full_geometry():
    """BASELINE configs[4]: ViT-MAE-L 480, 10-way 5-shot (51 images, 550 pairs) with the fp8 attention, at its own size.  The CPU oracle
    needs minutes here, so the fp8 run is held against the 16-bit run of the same model (itself pinned to <= 1e-3 on the fixtures and
    on the reduced cfg5-shaped episode above): <= the stated bound, finite, right shape."""
    lam, batch = _bench_model("cfg5")
    lam.selected_rows = torch.arange(batch["flag_examples"].shape[2])
    base = lam(batch)["logits"].clone()
    lam.attn_fp8 = True
    out8 = lam(batch)["logits"]
    assert out8.shape == base.shape == (1, 11, 480, 480) and bool(torch.isfinite(out8).all())
    e = rel_err(out8, base)
    print(f"[attn_fp8 cfg5] logits vs the 16-bit path: {e:.3e} (bound {FP8_BOUND})")
    assert 0 < e <= FP8_BOUND, e


@pytest.mark.parametrize("name", ["sam_tiny_2w2s_all_prompts", "hf_tiny_1w1s_masks", "cfg2_sam_b_1024_1w1s"])
def test_attention_rows_path_matches_the_v_transposed_path(name):
    """LamEngine.attn_rows (default): no V^T copies, no window buffers - la_attn_fwd_rows on image-order q | k | v.  The attention outputs are
    bit-identical to the V^T path's (tests/test_ops_gpu.py); what differs is the ORDER in which the token means of the proj operand are
    summed (image order instead of window order), i.e. fp32 rounding of the mean corrections: the logits agree to 1e-5."""
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    outs = []
    for rows in (True, False):
        lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
        lam.selected_rows = gold.get("selected_rows")
        lam.norm_fold = False              # (the folded-LayerNorm block stack exists for the rows path only: compare like with like)
        lam.engine().attn_rows = rows
        # the window blocks' token means from la_colmean16 on both sides (the V^T path has no other form): the attention epilogue's column
        # sums (LamEngine.win_fused_cs, the rows path's default) are the same means to 1e-7 in another summation order, and a last-bit
        # difference of a mean flips 16-bit roundings downstream - 2 - 4e-4 on the logits, "another draw" like any re-ordering
        lam.engine().win_fused_cs = False
        outs.append(lam(batch)["logits"].float().clone())
        del lam
        torch.cuda.empty_cache()
    err = rel_err(outs[0], outs[1])
    print(f"[attn_rows {name}] logits, rows path vs V^T path: {err:.3e}")
    assert err <= 1e-5


@pytest.mark.parametrize("name", ["cfg2_sam_b_1024_1w1s", "cfg1_mae_b_480_1w1s"])
def test_folded_layernorm_block_stack_against_the_layernorm_kernels(name):
    """LamEngine.norm_fold (round 6, the default of the wide fp16 encoders): norm1 / norm2 folded into the GEMMs on both sides
    (image_encoder.py:181-197).  Same mathematics, another realisation of the 16-bit roundings (rn16(W diag(gamma)) instead of rn16(W),
    rn16(x) instead of rn16(LayerNorm(x))): both forms hold the 1e-3 tolerance against the reference fixture on their own
    (test_episode_matches_reference_fixture runs the default; this test runs the other one) and they differ from each other by less than
    the sum of their errors."""
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    ref_logits = reference_logits(case, gold, batch)
    outs = []
    for fold in (True, False):
        lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
        lam.norm_fold = fold
        outs.append(lam(batch)["logits"].float().clone())
        assert lam.engine().norm_fold == fold
        del lam
        torch.cuda.empty_cache()
    e_fold, e_plain, e_between = rel_err(outs[0], ref_logits), rel_err(outs[1], ref_logits), rel_err(outs[0], outs[1])
    print(f"[norm_fold {name}] logits vs reference: folded {e_fold:.3e}, LayerNorm kernels {e_plain:.3e}; folded vs LayerNorm kernels {e_between:.3e}")
    assert e_fold <= 1e-3 and e_plain <= 1e-3 and 0 < e_between <= 2e-3


@pytest.mark.parametrize("size,folded", [(224, True), (96, False)])
def test_wide_hf_encoder_fold_dispatch_by_grid_size(size, folded):
    """A ViT-B-wide HF stack (two layers) at 224 px (197 tokens per image: the folded form with per-image groups that end inside row tiles,
    three images in one launch) and at 96 px (37 tokens: below the 128 rows a group of the producer epilogue needs - the engine keeps the
    LayerNorm kernels instead of failing): both against the oracle."""
    from labelanything_amd.config import EncoderSpec, LamConfig, register_encoder
    register_encoder("hf_wide2", EncoderSpec("hf", dim=768, depth=2, heads=12, mlp=3072, img_size=224))
    cfg = LamConfig(encoder="hf_wide2", image_size=size, image_embed_dim=768, embed_dim=64, spatial_convs=3, custom_preprocess=False)
    calls = []
    from labelanything_amd import _lib as L
    orig = L.norm_finalize
    L.norm_finalize = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        assert _encoder_vs_oracle(cfg, 3, 31) <= 1e-3
    finally:
        L.norm_finalize = orig
    assert bool(calls) == folded
