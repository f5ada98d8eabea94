"""Error budget of the 16-bit MFMA operand roundings of the image encoder (CPU emulation; diagnostic tool, not product).

The HIP encoder keeps an fp32 residual stream and fp32 accumulators; what it rounds to 16 bit are the MFMA OPERANDS:
the A side (LayerNorm output, q/k/v, softmax probabilities, attention output, GELU output, im2col patches) and the B side
(weights, rel-pos tables).  This tool re-runs the oracle's encoder on CPU in fp32 with exactly those roundings inserted
(``x.half().float()``), one group at a time or all but one, and reports max|d|/max|ref| on the embeddings and on the
low-res logits of the full episode, against the un-rounded oracle.  It is how the split-precision groups of
``LamEngine(precise=...)`` were chosen.

    python tools/error_budget.py sam_tiny_2w2s_all_prompts [--dtype f16|bf16] [--only GROUPS] [--loo] [--split GROUPS]

Modes per rounding point: "r" = rounded to 16 bit, "x" = exact, "s" = hi/lo split (hi = r16(x), lo = r16(x - hi); both
kept, i.e. ~22 mantissa bits), and for weights "m" = rounded, with the product of the TOKEN MEAN of the A operand (per image /
window) and the lo plane added back (--mean): the part of the weight-rounding error that is the same for every token.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from labelanything_amd.episodes import make_episode  # noqa: E402
from labelanything_amd.weights import init_state_dict  # noqa: E402
from oracle import lam_oracle as O  # noqa: E402
from tests.cases import CASES, geometry_for  # noqa: E402

# rounding points (A = activation operand, W = weight operand)
POINTS = ["patch.A", "patch.W", "qkv.A", "qk.W", "v.W", "attn.QK", "attn.V", "attn.T", "attn.P", "proj.A", "proj.W",
          "lin1.A", "lin1.W", "lin2.A", "lin2.W", "neck.A", "neck.W"]


class Policy:
    def __init__(self, dtype, modes, fold=False):
        self.fold = fold
        self.dt = dtype
        self.modes = modes          # point -> "r" | "x" | "s"   (optionally "point@block" overrides)

    def r16(self, x):
        return x.to(self.dt).float()

    def __call__(self, point, x, block=None):
        m = self.modes.get(f"{point}@{block}", self.modes.get(point, "x"))
        if m == "x":
            return x
        hi = self.r16(x)
        if m == "r":
            return hi
        return hi + self.r16(x - hi)


MEAN_GROUP = {"n": 0, "tokens": 0}       # SAM window blocks: windows per image / real tokens per image (set by sam_encoder)


def mean_corrected(pol, a, wt, bias):
    """a W_hi^T + mean_tokens(a) W_lo^T + bias (a: [batch, tokens..., K]).  In SAM window blocks the batch entries are windows
    (zero-padded): the mean is taken over the REAL tokens of the whole image, as a device implementation would (one column mean
    of the LayerNorm output / attention output per image)."""
    hi = pol.r16(wt)
    lo = pol.r16(wt - hi)
    dims = tuple(range(1, a.dim() - 1))
    g = MEAN_GROUP["n"]
    if g > 1 and a.shape[0] % g == 0:
        k = a.shape[-1]
        am = a.reshape(a.shape[0] // g, -1, k).sum(dim=1, keepdim=True) / MEAN_GROUP["tokens"]        # [images, 1, K]
        am = am.repeat_interleave(g, dim=0).view(a.shape[0], *([1] * len(dims)), k)
    else:
        am = a.mean(dim=dims, keepdim=True)
    return F.linear(a, hi, bias) + F.linear(am, lo)


def lin(pol, point, w, name, x, block=None):
    if pol.modes.get(point + ".W") == "m":
        return mean_corrected(pol, pol(point + ".A", x, block), w[name + ".weight"], w.get(name + ".bias"))
    wt = pol(point + ".W", w[name + ".weight"], block)
    return F.linear(pol(point + ".A", x, block), wt, w.get(name + ".bias"))


def fold_linear(pol, apoint, wpoint, wt, bias, gamma, beta, eps, x, blk):
    """LayerNorm folded into the consumer GEMM (--fold): the MFMA operand is the 16-bit rounding of the UN-normalised stream x, the
    weight is rn16(W diag(gamma)); the epilogue applies rstd (acc - mu c) + b' with c = row sums of the rounded plane, b' = b + W beta,
    mu / rstd the fp32 row statistics.  Weight mode "m": + mean_tokens(rstd (x16 - mu)) . W'_lo^T per image."""
    mu = x.mean(-1, keepdim=True)
    rstd = (x.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
    xa = pol(apoint, x, blk)
    wf = wt * gamma
    m = pol.modes.get(wpoint, "x")
    hi = wf if m == "x" else pol.r16(wf)
    if m == "s":
        hi = hi + pol.r16(wf - hi)
    y = rstd * (F.linear(xa, hi) - mu * hi.sum(1)) + (bias + wt @ beta)
    if m == "m":
        lo = pol.r16(wf - hi)
        z = rstd * (xa - mu)
        y = y + F.linear(z.mean(dim=tuple(range(1, z.dim() - 1)), keepdim=True), lo)
    return y


def sam_qkv_folded(pol, w, bp, x, blk, win):
    """q | k | v of one SAM block from the un-normalised stream (image order), window-partitioned with bias rows for padded tokens."""
    pre = bp + ".attn"
    e = x.shape[-1]
    wq, bq = w[pre + ".qkv.weight"], w[pre + ".qkv.bias"]
    g, b = w[bp + ".norm1.weight"], w[bp + ".norm1.bias"]
    qk = fold_linear(pol, "qkv.A", "qk.W", wq[:2 * e], bq[:2 * e], g, b, 1e-6, x, blk)
    v = fold_linear(pol, "qkv.A", "v.W", wq[2 * e:], bq[2 * e:], g, b, 1e-6, x, blk)
    qkv = torch.cat([qk, v], dim=-1)
    if win > 0:
        qkv, padded = O.window_split(qkv - bq, win)
        return qkv + bq, padded
    return qkv, None


def sam_attention(pol, w, pre, x, heads, blk, qkv=None):
    n, gh, gw = x.shape[:3]
    e = x.shape[-1] if qkv is None else x.shape[-1] // 3
    hd = e // heads
    t = gh * gw
    wq = w[pre + ".qkv.weight"]
    if qkv is not None:
        qkv = x.reshape(n, t, 3 * e)
    elif pol.modes.get("v.W") == "m":
        xa = pol("qkv.A", x.reshape(n, t, e), blk)
        qkv = torch.cat([F.linear(xa, pol("qk.W", wq[:2 * e], blk), w[pre + ".qkv.bias"][:2 * e]),
                         mean_corrected(pol, xa, wq[2 * e:], w[pre + ".qkv.bias"][2 * e:])], dim=-1)
    else:
        xa = pol("qkv.A", x.reshape(n, t, e), blk)
        wq = torch.cat([pol("qk.W", wq[:2 * e], blk), pol("v.W", wq[2 * e:], blk)])
        qkv = F.linear(xa, wq, w[pre + ".qkv.bias"])
    qkv = qkv.view(n, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = pol("attn.QK", qkv[0], blk), pol("attn.QK", qkv[1], blk), pol("attn.V", qkv[2], blk)
    scores = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    rh = pol("attn.T", O.rel_pos_table(gh, gh, w[pre + ".rel_pos_h"]), blk)
    rw = pol("attn.T", O.rel_pos_table(gw, gw, w[pre + ".rel_pos_w"]), blk)
    qg = q.reshape(n, heads, gh, gw, hd)
    bias_h = torch.einsum("nhyxc,ykc->nhyxk", qg, rh)
    bias_w = torch.einsum("nhyxc,xkc->nhyxk", qg, rw)
    scores = scores.view(n, heads, gh, gw, gh, gw) + bias_h[..., :, None] + bias_w[..., None, :]
    scores = scores.view(n, heads, t, t)
    # flash form: P = exp(s - max) rounded to 16 bit for the PV MFMA, the row sum taken from the UN-rounded fp32 values
    mx = scores.max(dim=-1, keepdim=True).values
    pe = torch.exp(scores - mx)
    o = (pol("attn.P", pe, blk) @ v) / pe.sum(dim=-1, keepdim=True)
    o = o.transpose(1, 2).reshape(n, gh, gw, e)
    return lin(pol, "proj", w, pre + ".proj", o, blk)


def patch_embed(pol, images, wt, bias, patch):
    """Patch embedding conv (stride = kernel): with patch.W in mode "m" one 16-bit weight plane + the product of the image's MEAN patch
    and the lo plane."""
    a = pol("patch.A", images)
    if pol.modes.get("patch.W") != "m":
        return F.conv2d(a, pol("patch.W", wt), bias, stride=patch)
    hi = pol.r16(wt)
    lo = pol.r16(wt - hi)
    y = F.conv2d(a, hi, bias, stride=patch)
    n, c, hh, ww = a.shape
    mp = a.view(n, c, hh // patch, patch, ww // patch, patch).mean(dim=(2, 4))           # mean patch per image [n, c, p, p]
    return y + torch.einsum("ncyx,ocyx->no", mp, lo)[:, :, None, None]


def sam_encoder(pol, w, geo, images, pre="image_encoder"):
    x = patch_embed(pol, images, w[pre + ".patch_embed.proj.weight"], w[pre + ".patch_embed.proj.bias"], geo.patch).permute(0, 2, 3, 1)
    x = x + w[pre + ".pos_embed"]
    for i in range(geo.enc_depth):
        bp = f"{pre}.blocks.{i}"
        win = 0 if i in geo.global_idx else geo.window
        if pol.fold:
            h, wd, nimg = x.shape[1], x.shape[2], x.shape[0]
            qkv, padded = sam_qkv_folded(pol, w, bp, x, i, win)
            if win > 0:
                MEAN_GROUP.update(n=qkv.shape[0] // nimg, tokens=h * wd)
            y = sam_attention(pol, w, bp + ".attn", qkv, geo.enc_heads, i, qkv=True)
            MEAN_GROUP.update(n=0)
            if win > 0:
                y = O.window_merge(y, win, padded, (h, wd))
            x = x + y
            z = O.gelu(fold_linear(pol, "lin1.A", "lin1.W", w[bp + ".mlp.lin1.weight"], w[bp + ".mlp.lin1.bias"], w[bp + ".norm2.weight"],
                                   w[bp + ".norm2.bias"], 1e-6, x, i))
            x = x + lin(pol, "lin2", w, bp + ".mlp.lin2", z, i)
            continue
        y = O.layer_norm(w, bp + ".norm1", x, 1e-6)
        if win > 0:
            h, wd = y.shape[1], y.shape[2]
            nimg = y.shape[0]
            y, padded = O.window_split(y, win)
            MEAN_GROUP.update(n=y.shape[0] // nimg, tokens=h * wd)
            y = sam_attention(pol, w, bp + ".attn", y, geo.enc_heads, i)
            MEAN_GROUP.update(n=0)
            y = O.window_merge(y, win, padded, (h, wd))
        else:
            y = sam_attention(pol, w, bp + ".attn", y, geo.enc_heads, i)
        x = x + y
        z = O.layer_norm(w, bp + ".norm2", x, 1e-6)
        z = lin(pol, "lin2", w, bp + ".mlp.lin2", O.gelu(lin(pol, "lin1", w, bp + ".mlp.lin1", z, i)), i)
        x = x + z
    last = x.permute(0, 3, 1, 2)
    if not geo.sam_neck:
        return last
    return conv_neck(pol, w, pre + ".neck", last)


def conv_neck(pol, w, pre, x):
    x = F.conv2d(pol("neck.A", x), pol("neck.W", w[pre + ".0.weight"]))
    x = O.layer_norm_2d(w, pre + ".1", x)
    x = F.conv2d(pol("neck.A", x), pol("neck.W", w[pre + ".2.weight"]), padding=1)
    return O.layer_norm_2d(w, pre + ".3", x)


def hf_encoder(pol, w, geo, images, pre="image_encoder"):
    bn = images.shape[0]
    g = images.shape[-1] // geo.patch
    e, heads = geo.enc_dim, geo.enc_heads
    hd = e // heads
    x = patch_embed(pol, images, w[pre + ".embeddings.patch_embeddings.projection.weight"],
                    w[pre + ".embeddings.patch_embeddings.projection.bias"], geo.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([w[pre + ".embeddings.cls_token"].expand(bn, -1, -1), x], dim=1)
    x = x + O.hf_pos_embed(w, pre, g, geo.hf_pos_grid)
    t = x.shape[1]
    for i in range(geo.enc_depth):
        lp = f"{pre}.encoder.layer.{i}"
        y = O.layer_norm(w, lp + ".layernorm_before", x, 1e-12)
        ya = pol("qkv.A", y, i)
        if pol.fold:
            q, k, v = (fold_linear(pol, "qkv.A", "v.W" if n == "value" else "qk.W", w[f"{lp}.attention.attention.{n}.weight"],
                                   w[f"{lp}.attention.attention.{n}.bias"], w[lp + ".layernorm_before.weight"], w[lp + ".layernorm_before.bias"],
                                   1e-12, x, i).view(bn, t, heads, hd).transpose(1, 2) for n in ("query", "key", "value"))
        else:
          q, k, v = ((mean_corrected(pol, ya, w[f"{lp}.attention.attention.{n}.weight"], w[f"{lp}.attention.attention.{n}.bias"])
                    if (n == "value" and pol.modes.get("v.W") == "m") else
                    F.linear(ya, pol("v.W" if n == "value" else "qk.W", w[f"{lp}.attention.attention.{n}.weight"], i),
                             w[f"{lp}.attention.attention.{n}.bias"])).view(bn, t, heads, hd).transpose(1, 2)
                   for n in ("query", "key", "value"))
        q, k, v = pol("attn.QK", q, i), pol("attn.QK", k, i), pol("attn.V", v, i)
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        mx = scores.max(dim=-1, keepdim=True).values
        pe = torch.exp(scores - mx)
        o = (pol("attn.P", pe, i) @ v) / pe.sum(dim=-1, keepdim=True)
        o = o.transpose(1, 2).reshape(bn, t, e)
        x = x + lin(pol, "proj", w, lp + ".attention.output.dense", o, i)
        if pol.fold:
            y = O.gelu(fold_linear(pol, "lin1.A", "lin1.W", w[lp + ".intermediate.dense.weight"], w[lp + ".intermediate.dense.bias"],
                                   w[lp + ".layernorm_after.weight"], w[lp + ".layernorm_after.bias"], 1e-12, x, i))
        else:
            y = O.layer_norm(w, lp + ".layernorm_after", x, 1e-12)
            y = O.gelu(lin(pol, "lin1", w, lp + ".intermediate.dense", y, i))
        x = x + lin(pol, "lin2", w, lp + ".output.dense", y, i)
    x = O.layer_norm(w, pre + ".layernorm", x, 1e-12)
    return x[:, 1:, :].reshape(bn, g, g, e).permute(0, 3, 1, 2).contiguous()


def run(pol, w, geo, batch, rows):
    """Oracle forward with the emulated encoder; returns (embeddings, low_res_logits, logits)."""
    if "images" in batch:
        im = batch["images"]
        b, n = im.shape[:2]
        enc = sam_encoder if geo.encoder == "sam" else hf_encoder
        e = enc(pol, w, geo, im.flatten(0, 1))
    else:
        e = batch["embeddings"]
        b, n = e.shape[:2]
        e = e.flatten(0, 1)
    if geo.lam_neck:
        e = conv_neck(pol, w, "neck", e)
    emb = e.view(b, n, *e.shape[1:])
    pts, bxs, msk = O.select_prompts(batch)
    pe = O.prompt_encoder(w, geo, emb[:, 1:], pts, bxs, msk, batch["flag_examples"], rows)
    low = O.mask_decoder(w, geo, emb[:, 0], pe["class_embeddings"])
    return emb, low, pe["class_embeddings"]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def rms(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--one-in", action="store_true", help="round ONE point at a time")
    ap.add_argument("--loo", action="store_true", help="round all points but one")
    ap.add_argument("--split", default="", help="comma list of points kept as hi/lo pairs (others rounded)")
    ap.add_argument("--exact", default="", help="comma list of points kept exact (others rounded)")
    ap.add_argument("--mean", default="", help="comma list of WEIGHT points (v.W, proj.W, lin1.W, lin2.W) rounded + token-mean correction")
    ap.add_argument("--fold", action="store_true", help="norm1 / norm2 folded into the qkv / lin1 GEMMs (operand = the un-normalised stream)")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    case = CASES[a.case]
    cfg = case["cfg"]
    geo = geometry_for(cfg)
    w = init_state_dict(cfg, seed=case["weight_seed"])
    batch = make_episode(**case["episode"])
    rows = None
    if cfg.bank_size:
        from tests.helpers import load_golden
        rows = load_golden(a.case)[0].get("selected_rows")

    def report(tag, modes):
        t0 = time.time()
        emb, low, cls = run(Policy(dt, modes, a.fold), w, geo, batch, rows)
        print(f"{tag:34s} emb {rel(emb, ref[0]):.3e} (rms {rms(emb, ref[0]):.3e})  cls {rel(cls, ref[2]):.3e}  "
              f"low {rel(low, ref[1]):.3e} (rms {rms(low, ref[1]):.3e})   [{time.time() - t0:.1f}s]", flush=True)

    with torch.no_grad():
        t0 = time.time()
        ref = run(Policy(dt, {}), w, geo, batch, rows)
        print(f"reference run {time.time() - t0:.1f}s")
        allr = {p: "r" for p in POINTS}
        modes = dict(allr)
        for p in filter(None, a.split.split(",")):
            modes[p] = "s"
        for p in filter(None, a.exact.split(",")):
            modes[p] = "x"
        for p in filter(None, a.mean.split(",")):
            modes[p] = "m"
        report("all rounded" if modes == allr else f"split={a.split} exact={a.exact} mean={a.mean}", modes)
        if a.one_in:
            for p in POINTS:
                report("only " + p, {p: "r"})
        if a.loo:
            for p in POINTS:
                m = dict(allr)
                m[p] = "x"
                report("all but " + p, m)


if __name__ == "__main__":
    with torch.no_grad():
        main()
