"""GPU: the training step (forward -> focal objective -> backward -> AdamW) of the HIP path against torch autograd of the CPU
oracle (oracle/lam_oracle.py + oracle/loss_oracle.py, both pinned on the reference) and against the reference's own parameter
update (tests/golden/train_step_*.safetensors, tools/make_golden_train.py)."""
import os

import pytest
import torch

from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from labelanything_amd.train import LamTrainer
from labelanything_amd.weights import init_state_dict
from oracle import lam_oracle as O
from oracle import loss_oracle as LO
from tests.cases import CASES, geometry_for
from tests.helpers import GOLDEN, load_golden, rel_err

pytestmark = pytest.mark.gpu


def make_gt(batch, n_classes, seed=0):
    """A ground-truth map per episode at the padded output size: random blocky labels, some ignored pixels."""
    g = torch.Generator().manual_seed(seed)
    dims = batch["dims"]
    b = dims.shape[0]
    hmax, wmax = int(dims[..., 0].max()), int(dims[..., 1].max())
    gt = torch.randint(0, n_classes, (b, (hmax + 15) // 16, (wmax + 15) // 16), generator=g)
    gt = gt.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :hmax, :wmax].contiguous()
    for i in range(b):
        h, w = int(dims[i, 0, 0]), int(dims[i, 0, 1])
        gt[i, h:, :] = -100
        gt[i, :, w:] = -100
    gt[torch.rand(gt.shape, generator=g) < 0.02] = -100
    return gt


def oracle_grads(case, batch, gt, rows, dtype=torch.float32):
    """Loss, logits and decoder-side gradients of the CPU oracle's autograd.  dtype=float64: a reference whose own accumulation error
    is negligible (sums over 10^5 pixels x 10^2 pairs cancel heavily: two correct fp32 implementations differ by ~1e-3 there)."""
    cfg = case["cfg"]
    w = {k: (v.to(dtype).requires_grad_("image_encoder" not in k and "gaussian" not in k) if v.is_floating_point() else v)
         for k, v in init_state_dict(cfg, case["weight_seed"]).items()}
    b = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    out = O.lam_forward(w, geometry_for(cfg), b, selected_rows=rows)
    loss, _ = LO.focal_objective(out["logits"], gt)
    loss.backward()
    return (float(loss.detach()), out["logits"].detach().float(),
            {k: v.grad.float() for k, v in w.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None})


@pytest.mark.parametrize("name", ["novit_d256_2w3s", "novit_d512_neck_1w2s", "sam_tiny_2w2s_all_prompts", "hf_tiny_1w1s_masks"])
def test_gradients_match_oracle_autograd(name):
    case = CASES[name]
    gold, _ = load_golden(name)
    batch = make_episode(**case["episode"])
    c = batch["flag_examples"].shape[2]
    gt = make_gt(batch, c, seed=3)
    rows = gold.get("selected_rows")
    ref_loss, ref_logits, ref_g = oracle_grads(case, batch, gt, rows)
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.selected_rows = rows
    tr = LamTrainer(lam)
    tr.zero_grad()
    res = tr.forward_backward(batch, gt)
    torch.cuda.synchronize()
    # forward of the training graph == inference engine == oracle (the encoder, when present, runs with 16-bit operands)
    tol_fwd = 1e-3 if case["cfg"].encoder_spec is not None else 2e-5
    assert rel_err(res["logits"], ref_logits) <= tol_fwd
    assert abs(float(res["loss"]) - ref_loss) <= tol_fwd * max(1.0, abs(ref_loss))
    # per-tensor max error relative to the tensor's own gradient scale, floored at 1e-2 of the largest gradient of the model:
    # several parameters have an analytically ZERO gradient (a key-projection bias shifts all scores of a softmax row alike, the
    # last class_mlp bias shifts all class logits of a pixel alike) and the reference's value for them is rounding noise
    gmax = max(float(v.abs().max()) for v in ref_g.values())
    worst = {}
    for k, gv in zip(tr.names, tr.opt.grad_views):
        ref = ref_g.get(k)
        if ref is None:                 # never reached by the forward (dead final attention of the prompt encoder's transformer)
            assert float(gv.abs().max()) == 0.0, k
            continue
        assert torch.isfinite(gv).all(), k
        worst[k] = float((gv.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-2 * gmax)
    # Decoder-only models: the gradient must match to accumulation accuracy.  With an image encoder in front, the embeddings carry
    # the 16-bit operand error of the HIP encoder (3-6e-4) and the gradient of a random-weight decoder amplifies that by up to two
    # orders of magnitude (measured 2-6e-2 on hf_tiny; with the ORACLE's embeddings fed in instead the same graph is at 2.4e-4:
    # test_decoder_graph_is_exact_behind_an_encoder below) - the image path is therefore only held to 1e-1 here.
    tol = 1e-1 if case["cfg"].encoder_spec is not None else 3e-4
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    assert set(k for k in ref_g) <= set(tr.names)


def test_decoder_graph_is_exact_behind_an_encoder():
    """hf_tiny with the oracle's (fp32) encoder output fed in as precomputed pre-neck embeddings: LAM neck + decoder gradients
    match the oracle's autograd to accumulation accuracy, i.e. the looser bound of the image path above is the encoder's
    16-bit operands, not the training graph."""
    case = CASES["hf_tiny_1w1s_masks"]
    batch = make_episode(**case["episode"])
    gt = make_gt(batch, batch["flag_examples"].shape[2], seed=3)
    ref_loss, _, ref_g = oracle_grads(case, batch, gt, None)
    w = init_state_dict(case["cfg"], case["weight_seed"])
    with torch.no_grad():
        im = batch["images"]
        b, n = im.shape[:2]
        e = O.encode_images(w, geometry_for(case["cfg"]), im.flatten(0, 1))
    b2 = {k: v for k, v in batch.items() if k != "images"}
    b2["embeddings"] = e.view(b, n, *e.shape[1:])
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    tr = LamTrainer(lam)
    tr.zero_grad()
    res = tr.forward_backward(b2, gt)
    assert abs(float(res["loss"]) - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss))
    gmax = max(float(v.abs().max()) for v in ref_g.values())
    for k, gv in zip(tr.names, tr.opt.grad_views):
        if k in ref_g:
            assert float((gv.cpu() - ref_g[k]).abs().max()) / max(float(ref_g[k].abs().max()), 1e-2 * gmax) <= 4e-4, k


def test_cfg3_train_step_at_full_size():
    """BASELINE cfg3 at its own size - ViT-MAE-B 480 px, 5-way 5-shot, one episode (26 images, 150 prompt pairs, 900 tokens each):
    the whole training step (encoder forward, decoder graph, focal objective, backward) runs through the HIP path from the IMAGES;
    the oracle's autograd gets the same episode with the HIP encoder's output as precomputed pre-neck embeddings (the encoder is the
    frozen part of this trainer), so loss and every decoder-side gradient must agree to accumulation accuracy."""
    import bench
    from labelanything_amd.config import LamConfig
    wl = bench.WORKLOADS["cfg3_train"]
    cfg = LamConfig(**wl["model"])
    batch = make_episode(batch=1, seed=31, prompts=("mask", "point"), **wl["episode"])
    c = batch["flag_examples"].shape[2]
    gt = make_gt(batch, c, seed=5)
    rows = torch.tensor([3, 14, 15, 92, 65, 35])
    lam = Lam(cfg, seed=5).cuda()
    lam.selected_rows = rows
    im = batch["images"]
    b, n = im.shape[:2]
    e = lam.image_encoder(im.flatten(0, 1).cuda()).float().cpu()                  # (26, 768, 30, 30) pre-neck
    tr = LamTrainer(lam)
    tr.zero_grad()
    res = tr.forward_backward(batch, gt)
    torch.cuda.synchronize()
    b2 = {k: v for k, v in batch.items() if k != "images"}
    b2["embeddings"] = e.view(b, n, *e.shape[1:])
    case = {"cfg": cfg, "weight_seed": 5}
    ref_loss, ref_logits, ref_g = oracle_grads(case, b2, gt, rows, dtype=torch.float64)
    assert res["logits"].shape == (1, 6, 480, 480)
    e_log = rel_err(res["logits"], ref_logits)
    assert e_log <= 5e-5, e_log
    assert abs(float(res["loss"]) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (float(res["loss"]), ref_loss)
    # The reference here is the oracle's autograd in FLOAT64: two correct fp32 implementations differ by more than either's error (the
    # fp32 oracle itself is 6e-3 away).  The gradients are sums over 2.46 M mask pixels (150 pairs x 128 x 128) or 230 400 output
    # pixels that cancel heavily; every per-pixel term is an fp32 value, so the sums carry ~ sqrt(N) eps of the terms' scale whatever
    # the order of accumulation: against float64 the HIP path leaves an ABSOLUTE error of 0.6 - 1.7e-4 of the model's largest gradient
    # entry (tools/train_grad_diag.py: the same tensors on every run, 1e-5 of it moves with the order of the atomics; the worst is
    # mask_downscaling.0.bias - itself the largest entry - then the last spatial convs).  Tensors with entries within 10x of the
    # largest therefore stay below 1e-3 relative; analytically-zero ones (last class-MLP bias) are all error.
    gmax = max(float(v.abs().max()) for v in ref_g.values())
    by_name = dict(zip(tr.names, tr.opt.grad_views))
    abs_err, rel_own = {}, {}
    for k, ref in ref_g.items():
        err = float((by_name[k].cpu() - ref).abs().max())
        abs_err[k] = err / gmax
        rel_own[k] = err / max(float(ref.abs().max()), 1e-3 * gmax)
    wa, wr = max(abs_err, key=abs_err.get), max(rel_own, key=rel_own.get)
    print(f"cfg3 full-size training step: logits {e_log:.2e}; worst absolute gradient error {abs_err[wa]:.2e} of the largest entry ({wa}), "
          f"worst error relative to a tensor's own scale {rel_own[wr]:.2e} ({wr})")
    assert len(ref_g) >= 150
    assert abs_err[wa] <= 4e-4, (wa, abs_err[wa])
    big = {k: v for k, v in rel_own.items() if float(ref_g[k].abs().max()) >= 0.1 * gmax}
    assert big and max(big.values()) <= 1e-3, sorted(big.items(), key=lambda kv: -kv[1])[:4]


def test_three_steps_match_the_reference_fixture():
    """tests/golden/train_step.safetensors = the REFERENCE's WrapperModule + LabelAnythingLoss + torch AdamW + HF warm-up schedule
    (tools/make_golden_train.py): losses of three steps, the first gradient and the total parameter change of every tensor."""
    import json
    from safetensors.torch import load_file
    from tests.cases import TRAIN_CASE as case
    gold = load_file(os.path.join(GOLDEN, "train_step.safetensors"))
    with open(os.path.join(GOLDEN, "train_step.json")) as fh:
        keys = json.load(fh)["keys"]
    batch = make_episode(**case["episode"])
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.selected_rows = gold["selected_rows"]
    start = {k: p.detach().clone() for k, p in lam.named_parameters()}
    tr = LamTrainer(lam, lr=case["lr"], weight_decay=case["weight_decay"], num_warmup_steps=case["warmup"])
    assert sorted(tr.names) == keys
    losses = []
    for step in range(case["steps"]):
        tr.zero_grad()
        res = tr.forward_backward(batch, gold["gt"])
        losses.append(float(res["loss"]))
        if step == 0:
            assert rel_err(res["logits"], gold["logits0"]) <= 2e-5
            g0 = {k: gv.clone() for k, gv in zip(tr.names, tr.opt.grad_views)}
        tr.apply_update()
        lam.invalidate()
    assert torch.allclose(torch.tensor(losses), gold["loss"], rtol=2e-5, atol=0), (losses, gold["loss"])
    gn = torch.stack([g0[k].norm() for k in keys]).cpu()
    dn = torch.stack([(dict(lam.named_parameters())[k].detach() - start[k]).norm() for k in keys]).cpu()
    floor_g = 1e-3 * float(gold["grad_norm"].max())
    assert float(((gn - gold["grad_norm"]).abs() / gold["grad_norm"].clamp_min(floor_g)).max()) <= 1e-3
    # AdamW's step is lr * m / (sqrt(v) + eps): a tensor whose gradient is rounding noise (analytically zero: key-projection biases,
    # the last class_mlp bias) moves by ~lr per entry in a noise-given direction, in the reference as here - its total change is
    # only bounded; every tensor with a real gradient must move exactly like the reference's, and the tensors the forward never
    # reaches must not move at all (torch skips .grad = None tensors, weight decay included)
    real = ((gn - gold["grad_norm"]).abs() <= 1e-3 * gold["grad_norm"]) & (gold["grad_norm"] > 0)   # gradient is signal, not noise
    assert int(real.sum()) >= 0.8 * len(keys)
    dead = gold["delta_norm"] == 0           # .grad was None in the reference: AdamW skipped the tensor, weight decay included
    zero_g = (gold["grad_norm"] == 0) & ~dead  # a ZERO gradient tensor is decayed, here as there (point_embeddings.3: no real 2nd corner)
    assert float(((dn - gold["delta_norm"]).abs() / gold["delta_norm"].clamp_min(1e-12))[zero_g].max()) <= 1e-4 if bool(zero_g.any()) else True
    # (single ENTRIES of a real tensor can still be noise - units that are almost dead - so the norms agree tightly for most tensors
    # and loosely for all; the entry-wise check below is the sharp one)
    rel_d = ((dn - gold["delta_norm"]).abs() / gold["delta_norm"].clamp_min(1e-12))[real]
    assert float(rel_d.quantile(0.9)) <= 2e-3 and float(rel_d.max()) <= 0.5, (float(rel_d.quantile(0.9)), float(rel_d.max()))
    assert float(dn[dead].abs().max()) == 0.0 and int(dead.sum()) >= 10
    params = dict(lam.named_parameters())
    gmax = max(float(v.abs().max()) for k, v in gold.items() if k.startswith("grad."))
    for k, v in gold.items():
        if k.startswith("grad."):
            assert float((g0[k[5:]].cpu() - v).abs().max()) <= 3e-4 * max(float(v.abs().max()), 1e-2 * gmax), k
        if k.startswith("final."):
            name = k[6:]
            mine, ref0 = params[name].detach().cpu(), start[name].cpu()
            sig = gold["grad." + name].abs() > 1e-2 * gold["grad." + name].abs().max()             # entries with a real gradient
            step_ref, step_mine = (v - ref0)[sig], (mine - ref0)[sig]
            assert float((step_mine - step_ref).abs().max()) <= 2e-2 * float(step_ref.abs().max()), name


def test_step_updates_weights_and_invalidates_the_engine():
    """FlatAdamW writes through raw pointers; the inference engine must re-pack afterwards."""
    case = CASES["novit_d512_neck_1w2s"]
    gold, _ = load_golden("novit_d512_neck_1w2s")
    batch = make_episode(**case["episode"])
    gt = make_gt(batch, batch["flag_examples"].shape[2], seed=4)
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.selected_rows = gold.get("selected_rows")
    before = lam(batch)["logits"].clone()
    tr = LamTrainer(lam, lr=1e-3)
    l0 = float(tr.step(batch, gt)["loss"])
    after = lam(batch)["logits"]
    assert not torch.allclose(before, after)                      # the forward sees the updated weights ...
    fresh = Lam(case["cfg"], seed=123).cuda()
    fresh.load_state_dict(lam.state_dict())
    fresh.selected_rows = lam.selected_rows
    assert rel_err(after, fresh(batch)["logits"]) < 1e-6          # ... exactly as a freshly built model with the same weights does
    losses = [l0] + [float(tr.step(batch, gt)["loss"]) for _ in range(5)]
    assert losses[-1] < losses[0]                                 # and the objective goes down on the batch it is trained on


@pytest.mark.parametrize("mnk", [(135000, 128, 256), (900, 256, 128), (150, 256, 2048), (300000, 4, 4), (70000, 7, 3), (4099, 33, 65),
                                 (300000, 256, 16), (300000, 16, 16), (100000, 16, 256), (50001, 132, 260), (4096, 256, 768)])
def test_weight_gradient_gemm_tn_matches_torch(mnk):
    """la_gemm_tn: dW += dY^T X on the 32x32x2 fp32 MFMA (row chunks folded with atomics; rows staged through LDS for long aligned
    reductions, in the 128 x 128 / 256 x 32 / 32 x 256 / 32 x 32 tile shapes) and, for tiny outputs over very long reductions, on the
    VALU; accumulation into a pre-loaded dW, the bias gradient on the way, operands that are column slices of wider matrices."""
    from labelanything_amd import _lib as L
    m, n, k = mnk
    g = torch.Generator().manual_seed(m + n)
    dy = torch.randn(m, n, generator=g).cuda()
    x = torch.randn(m, k, generator=g).cuda()
    dw0 = torch.randn(n, k, generator=g).cuda()
    dw = dw0.clone()
    L.gemm_tn(dy, x, dw)
    torch.cuda.synchronize()
    ref = dw0.double() + dy.double().t() @ x.double()
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # with the bias gradient, on column slices (row stride = the parent's width)
    wide_dy = torch.randn(m, n + 8, generator=g).cuda()
    wide_x = torch.randn(m, k + 4, generator=g).cuda()
    dys, xs = wide_dy[:, 4:4 + n], wide_x[:, 4:4 + k]
    dw, db = dw0.clone(), torch.ones(n, device="cuda")
    L.gemm_tn(dys, xs, dw, db)
    torch.cuda.synchronize()
    ref = dw0.double() + dys.double().t() @ xs.double()
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    refb = 1.0 + dys.double().sum(0)
    # (fp32 atomics fold the row chunks in arrival order: 300 000-row column sums measure up to 2.1e-6 box to box)
    assert float((db.double() - refb).abs().max() / refb.abs().max()) < 5e-6


@pytest.mark.parametrize("mn", [(135000, 128), (2457600, 4), (614400, 256), (150, 2048), (6, 256), (1000, 300)])
def test_bias_gradient_colsum_matches_torch(mn):
    """la_colsum_acc: out[n] += sum_m dY[m][n] (the bias gradient of every linear / conv layer)."""
    from labelanything_amd import _lib as L
    m, n = mn
    g = torch.Generator().manual_seed(m * 3 + n)
    dy = torch.randn(m, n, generator=g).cuda()
    out0 = torch.randn(n, generator=g).cuda()
    out = out0.clone()
    L.colsum_acc(dy, out)
    torch.cuda.synchronize()
    ref = out0.double() + dy.double().sum(0)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("rows_e", [(70000, 4), (70000, 16), (66000, 32), (5000, 16), (3000, 64), (2000, 256), (700, 512), (901, 768), (500, 1024), (300, 1280)])
def test_layernorm_backward_matches_torch_autograd(rows_e, gelu):
    """la_layernorm_bwd on every dispatch branch: one thread per row for the 4 / 16 / 32-channel LayerNorm2d stacks over >= 65536
    pixels (mask_downscaling, output_upscaling), a wave per row otherwise (decoder 256, encoder 768 / 1280)."""
    from labelanything_amd import autograd_ops as A
    rows, e = rows_e
    g = torch.Generator().manual_seed(rows + e)
    x = torch.randn(rows, e, generator=g).cuda().requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(e, generator=g)).cuda().requires_grad_(True)
    beta = (0.2 * torch.randn(e, generator=g)).cuda().requires_grad_(True)
    dy = torch.randn(rows, e, generator=g).cuda()
    y = A.layer_norm(x, gamma, beta, 1e-6, gelu)
    y.backward(dy)
    xr, gr, br = (t.detach().double().cpu().requires_grad_(True) for t in (x, gamma, beta))
    yr = torch.nn.functional.layer_norm(xr, (e,), gr, br, 1e-6)
    if gelu:
        yr = torch.nn.functional.gelu(yr)
    yr.backward(dy.double().cpu())
    assert rel_err(y.detach(), yr.detach().float()) < 1e-5          # (4 channels at eps 1e-6: the fp32 variance itself is 4e-6 away from fp64)
    for mine, ref in ((x.grad, xr.grad), (gamma.grad, gr.grad), (beta.grad, br.grad)):
        assert float((mine.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("shape", [(2, 150, 150, 8, 16), (3, 1, 900, 8, 16), (3, 900, 1, 8, 16), (2, 6, 6, 8, 32), (5, 6, 6, 8, 32), (2, 6, 900, 8, 16),
                                   (2, 900, 6, 8, 16), (2, 300, 40, 4, 64), (2, 7, 300, 2, 4), (1, 256, 256, 2, 8), (2, 520, 3, 8, 64)])
def test_small_attention_backward_matches_torch_autograd(shape):
    """autograd_ops.attention (la_attn_small + la_attn_small_lse + la_attn_small_bwd): both backward forms - the long side in threads, the
    short side's gradients added up per workgroup through LDS - against torch autograd in fp64."""
    from labelanything_amd import autograd_ops as A
    b, nq, nk, heads, hd = shape
    g = torch.Generator().manual_seed(nq * 7 + nk)
    e = heads * hd
    q = torch.randn(b * nq, e, generator=g).cuda().requires_grad_(True)
    k = torch.randn(b * nk, e, generator=g).cuda().requires_grad_(True)
    v = torch.randn(b * nk, e, generator=g).cuda().requires_grad_(True)
    w = torch.randn(b * nq, e, generator=g).cuda()
    o = A.attention(q, k, v, b, nq, nk, heads)
    (o * w).sum().backward()
    qd, kd, vd = (t.detach().double().cpu().requires_grad_(True) for t in (q, k, v))
    q4 = qd.view(b, nq, heads, hd).transpose(1, 2)
    k4 = kd.view(b, nk, heads, hd).transpose(1, 2)
    v4 = vd.view(b, nk, heads, hd).transpose(1, 2)
    ref = (torch.softmax(q4 @ k4.transpose(-1, -2) / hd ** 0.5, -1) @ v4).transpose(1, 2).reshape(b * nq, e)
    (ref * w.double().cpu()).sum().backward()
    assert float((o.detach().double().cpu() - ref.detach()).abs().max()) <= 2e-6 * float(ref.abs().max())
    for name, mine, r in (("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        # (one key: softmax == 1, dq and dk are exactly zero in the reference - the bound is relative to the data scale then)
        assert float((mine.double().cpu() - r).abs().max()) <= 5e-6 * max(1.0, float(r.abs().max())), name


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_layernorm_backward_with_skip_connection_and_16bit_copy(dt):
    """la_layernorm_bwd_res == la_layernorm_bwd followed by the skip connection's add and a cast (in place on the running gradient)."""
    from labelanything_amd import _lib as L
    rows, e = 901, 768
    g = torch.Generator().manual_seed(5)
    x, dy, skip = (torch.randn(rows, e, generator=g).cuda() for _ in range(3))
    gamma, beta = (1 + 0.3 * torch.randn(e, generator=g)).cuda(), torch.randn(e, generator=g).cuda()
    dx, dg0, db0 = torch.empty_like(x), torch.zeros(e, device="cuda"), torch.zeros(e, device="cuda")
    L.layernorm_bwd(x, dy, gamma, beta, 1e-6, False, dx, dg0, db0)
    run, out16 = skip.clone(), torch.empty(rows, e, device="cuda", dtype=dt)
    dg1, db1 = torch.zeros(e, device="cuda"), torch.zeros(e, device="cuda")
    L.layernorm_bwd_res(x, dy, gamma, beta, 1e-6, run, run, out16, dg1, db1)
    torch.cuda.synchronize()
    assert torch.equal(run, dx + skip)
    assert torch.equal(out16, (dx + skip).to(dt))
    assert float((dg1 - dg0).abs().max()) <= 1e-5 * float(dg0.abs().max()) and float((db1 - db0).abs().max()) <= 1e-5 * float(db0.abs().max())


def test_weight_transposes_in_one_launch_and_the_trainer_side_cache():
    """la_transpose_many on ragged shapes, and autograd_ops.WeightTransposes: first sight = own launch, afterwards one batched refresh per
    invalidate(); tensors outside the registered address range are never cached."""
    from labelanything_amd import _lib as L
    from labelanything_amd import autograd_ops as A
    g = torch.Generator().manual_seed(11)
    flat = torch.randn(33 * 70 + 256 * 256 + 5, generator=g).cuda()
    views = [flat[:33 * 70].view(33, 70), flat[33 * 70:33 * 70 + 65536].view(256, 256), flat[33 * 70 + 65536:].view(5, 1)]
    wt = A.WeightTransposes(enabled=True, lo=flat.data_ptr(), hi=flat.data_ptr() + 4 * flat.numel())
    first = [wt.get(v) for v in views]                       # first sight: one launch each, registered
    for v, t in zip(views, first):
        assert torch.equal(t, v.t().contiguous())
    flat.mul_(-2.0)                                          # "optimizer step"
    wt.invalidate()
    again = [wt.get(v) for v in views]                       # one la_transpose_many launch refreshes all three
    torch.cuda.synchronize()
    for v, t, f in zip(views, again, first):
        assert t.data_ptr() == f.data_ptr() and torch.equal(t, v.t().contiguous())
    other = torch.randn(7, 9, generator=g).cuda()            # not a view of the flat buffer: transposed on its own, not cached
    assert torch.equal(wt.get(other), other.t().contiguous()) and len(wt.entries) == 3


@pytest.mark.parametrize("period,reps,d,di", [(30, 7, 64, 32), (0, 7, 64, 32), (900, 3, 256, 128)])
def test_linear_fan_matches_separate_linear_nodes(period, reps, d, di):
    """autograd_ops.linear_fan (several projections of one input as ONE node: the image-side k / v / q of a two-way layer) against the
    separate ``add_rows`` + ``linear`` nodes it replaces: outputs bit-identical (same launches), data gradient equal to autograd's
    fan-in sum up to the order of three fp32 additions, weight / bias gradients bit-identical (same kernel on the same operands).
    The 2700-row case runs the data gradients on the large-M exact-fp32 MFMA kernel, whose residual epilogue accumulates IN PLACE
    (res = out32 = dx) - the form the real 270000 x 256 stream takes; 210 rows stay on the few-row kernel."""
    from labelanything_amd import autograd_ops as A
    g = torch.Generator().manual_seed(17)
    rows = (period or 30) * reps
    x0 = torch.randn(rows, d, generator=g).cuda()
    pe = torch.randn(period, d, generator=g).cuda() if period else None
    ws = [(torch.randn(di, d, generator=g) / 8).cuda() for _ in range(3)]
    bs = [torch.randn(di, generator=g).cuda(), None, torch.randn(di, generator=g).cuda()]
    use = [True, False, True]
    rs = [torch.randn(rows, di, generator=g).cuda() for _ in range(3)]

    def leaves():
        return (x0.clone().requires_grad_(True), [w.clone().requires_grad_(True) for w in ws],
                [b.clone().requires_grad_(True) if b is not None else None for b in bs])

    x, w, b = leaves()
    outs = A.linear_fan(x, pe, [(w[i], b[i], use[i] and pe is not None) for i in range(3)])
    sum((o * r).sum() for o, r in zip(outs, rs)).backward()
    x2, w2, b2 = leaves()
    xp = A.add_rows(x2, pe) if pe is not None else x2
    outs2 = [A.linear(xp if use[i] else x2, w2[i], b2[i]) for i in range(3)]
    sum((o * r).sum() for o, r in zip(outs2, rs)).backward()
    torch.cuda.synchronize()
    for o, o2 in zip(outs, outs2):
        assert torch.equal(o, o2)
    assert rel_err(x.grad, x2.grad) < 1e-6
    for i in range(3):
        if rows < 256:          # the register-fed weight-gradient kernel: one summation order
            assert torch.equal(w[i].grad, w2[i].grad)
            if b[i] is not None:
                assert torch.equal(b[i].grad, b2[i].grad)
        else:                   # >= 256 rows: row chunks meet in fp32 atomics - equal up to the order of the additions
            assert rel_err(w[i].grad, w2[i].grad) < 1e-6
            if b[i] is not None:
                assert rel_err(b[i].grad, b2[i].grad) < 1e-6
    if pe is not None:
        with pytest.raises(ValueError, match="period"):
            A.linear_fan(x0, pe[:-1], [(ws[0], bs[0], True)])
        with pytest.raises(ValueError, match="gradient"):
            A.linear_fan(x0, pe.clone().requires_grad_(True), [(ws[0], bs[0], True)])
