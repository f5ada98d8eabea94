#!/usr/bin/env python
"""fp8 (e4m3) QK^T attention, the opt-in switch for BASELINE configs[4] ("ViT-MAE-L 480px, N-way K=5 episodes, fp8 MFMA attention"):
kernel time next to the 16-bit kernel on the cfg5 / cfg3 / cfg1 attention shapes, and the logit error it costs on the cfg1 golden
fixture and on a cfg5-shaped episode (against the fp16 path; the fixture pins the fp16 path at < 1e-3)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L
from labelanything_amd.config import LamConfig
from labelanything_amd.episodes import make_episode
from labelanything_amd.models import Lam
from tests.cases import CASES
from tests.helpers import load_golden, reference_logits, rel_err


def bench(fn, it=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


print("kernel time, one layer's attention (microseconds; TFLOP/s on 4 T^2 64 FLOP per image-head)")
for name, b, heads, t in (("cfg5 MAE-L 480 (51 images x 16 heads, T 901)", 51, 16, 901), ("cfg3 MAE-B 480 (26 x 12, T 901)", 26, 12, 901),
                          ("cfg1 x32 episodes (64 x 12, T 901)", 64, 12, 901), ("long T (8 x 12, T 4096)", 8, 12, 4096)):
    e = heads * 64
    tpad = (t + 63) // 64 * 64
    qkv = (torch.randn(b * t, 3 * e, device="cuda") * 0.8).half()
    vt = torch.empty(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
    L.head_transpose(qkv, 2 * e, b, heads, t, tpad, vt)
    out = torch.empty(b * t, e, dtype=torch.float16, device="cuda")
    qk8 = torch.empty(b * t, 2 * e, dtype=torch.uint8, device="cuda")
    sc = 1 / math.sqrt(64)
    t16 = bench(lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t, tpad, 0, e, sc, L.ATTN_PLAIN))
    tcv = bench(lambda: L.qk_fp8(qkv, e, qk8))
    t8 = bench(lambda: L.attn_fwd_fp8(qk8, vt, out, b, heads, t, tpad, e, sc))
    fl = 4.0 * b * heads * t * t * 64
    print(f"  {name}: 16-bit {t16:8.1f} us ({fl / t16 / 1e6:6.0f} TF/s)   fp8 QK^T {t8:8.1f} us ({fl / t8 / 1e6:6.0f} TF/s) + {tcv:6.1f} us for the "
          f"e4m3 copy of q | k = {(t8 + tcv) / t16:.2f}x the 16-bit time")

print("logit error (max |a - b| / max |b|)")
case = CASES["cfg1_mae_b_480_1w1s"]
gold, _ = load_golden("cfg1_mae_b_480_1w1s")
batch = make_episode(**case["episode"])
ref = reference_logits(case, gold, batch)
for fp8 in (False, True):
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    lam.attn_fp8 = fp8
    out = lam(batch)["logits"]
    am = out.argmax(1).cpu()
    print(f"  cfg1 golden (reference fixture), attn_fp8={fp8}: logits {rel_err(out, ref):.3e}, argmax differs on {float((am != gold['argmax'].long()).float().mean()) * 100:.3f} % of the pixels")
cfg5 = LamConfig(encoder="vit_l_mae", image_size=480, image_embed_dim=1024, embed_dim=256, spatial_convs=3,
                 class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256}, custom_preprocess=False)
batch = make_episode(batch=1, seed=7, prompts=("mask",), n_ways=10, k_shots=5, image_size=480)
lam = Lam(cfg5, seed=3).cuda()
lam.selected_rows = torch.arange(11)
base = lam(batch)["logits"].clone()
lam.attn_fp8 = True
lam.invalidate()
out8 = lam(batch)["logits"]
print(f"  cfg5 MAE-L 480 10-way 5-shot episode, attn_fp8 vs the fp16 path: logits {rel_err(out8, base):.3e}, argmax differs on "
      f"{float((out8.argmax(1) != base.argmax(1)).float().mean()) * 100:.3f} % of the pixels")
