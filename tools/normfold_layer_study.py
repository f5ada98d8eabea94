#!/usr/bin/env python
"""CPU study behind LamEngine.norm_fold: layer by layer through the cfg1 HF stack (exact fp32 stream), the error that the 16-bit
rounding of the fc1 weight leaves on the pre-activation in the two forms - LayerNorm -> rn16(W) (the LayerNorm kernels) and the folded
rstd (x rn16(W gamma)^T - mean c) + b' - against the exact product, and the part of it that is shared by all tokens of an image.

    python tools/normfold_layer_study.py            (profiles/r06_normfold_layer_study.log)
"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from labelanything_amd.weights import init_state_dict
from labelanything_amd.episodes import make_episode
from tests.cases import CASES, geometry_for
from oracle import lam_oracle as O

torch.set_num_threads(int(os.environ.get("THREADS", 32)))
case = CASES["cfg1_mae_b_480_1w1s"]
cfg = case["cfg"]
geo = geometry_for(cfg)
w = init_state_dict(cfg, seed=case["weight_seed"])
im = make_episode(**case["episode"])["images"].flatten(0, 1)
pre = "image_encoder"
r16 = lambda t: t.half().float()
with torch.no_grad():
    x = F.conv2d(im, w[pre + ".embeddings.patch_embeddings.projection.weight"], w[pre + ".embeddings.patch_embeddings.projection.bias"],
                 stride=geo.patch).flatten(2).transpose(1, 2)
    x = torch.cat([w[pre + ".embeddings.cls_token"].expand(x.shape[0], -1, -1), x], 1) + O.hf_pos_embed(w, pre, im.shape[-1] // geo.patch, geo.hf_pos_grid)
    for i in range(geo.enc_depth):
        lp = f"{pre}.encoder.layer.{i}"
        y = O.layer_norm(w, lp + ".layernorm_before", x, 1e-12)
        bn, t, e = x.shape
        heads = geo.enc_heads
        hd = e // heads
        q, k, v = (F.linear(y, w[f"{lp}.attention.attention.{n}.weight"], w[f"{lp}.attention.attention.{n}.bias"]).view(bn, t, heads, hd).transpose(1, 2)
                   for n in ("query", "key", "value"))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(bn, t, e), w[lp + ".attention.output.dense.weight"], w[lp + ".attention.output.dense.bias"])
        g, b = w[lp + ".layernorm_after.weight"], w[lp + ".layernorm_after.bias"]
        W, bb = w[lp + ".intermediate.dense.weight"], w[lp + ".intermediate.dense.bias"]
        mu = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        rstd = (var + 1e-12).rsqrt()
        z = (x - mu) * rstd
        y = z * g + b
        h_exact = F.linear(y, W, bb)
        h_old = F.linear(y, r16(W), bb)
        Wf = r16(W * g)
        h_new = rstd * (F.linear(x, Wf) - mu * Wf.sum(1)) + (bb + W @ b)
        s = float(h_exact.abs().max())
        print(f"layer {i:2d}: |mean| / std of a row {float((mu.abs() / var.sqrt()).mean()):.3f} (max {float((mu.abs() / var.sqrt()).max()):.2f})  "
              f"rms of the token mean of z {float(z.mean(1).pow(2).mean().sqrt()):.3f}  fc1 pre-activation error: LayerNorm -> rn16(W) "
              f"{float((h_old - h_exact).abs().max()) / s:.2e}, folded {float((h_new - h_exact).abs().max()) / s:.2e};  token-shared part "
              f"{float((h_old - h_exact).mean(1).abs().max()) / s:.2e} / {float((h_new - h_exact).mean(1).abs().max()) / s:.2e}", flush=True)
        x = x + F.linear(O.gelu(h_exact), w[lp + ".output.dense.weight"], w[lp + ".output.dense.bias"])
