#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels on one MI355X (GEMM shapes of SAM ViT-B, attention)."""
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dt = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
    print("device", torch.cuda.get_device_name(0), "dtype", dt)
    for (m, n, k, name) in [(8192, 2304, 768, "qkv"), (8192, 768, 768, "proj"), (8192, 3072, 768, "lin1"),
                            (8192, 768, 3072, "lin2"), (32768, 2304, 768, "qkv x4"), (32768, 3072, 768, "lin1 x4"),
                            (32768, 768, 3072, "lin2 x4"), (65536, 768, 768, "proj x8"), (78400, 768, 768, "projw x8"),
                            (65536, 768, 3072, "lin2 x8"), (65536, 3072, 768, "lin1 x8"), (4096, 4096, 4096, "4k cube")]:
        a = torch.randn(m, k, device="cuda").to(dt)
        w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).to(dt)
        bias = torch.randn(n, device="cuda")
        o16 = torch.empty(m, n, device="cuda", dtype=dt)
        t = timeit(lambda: L.gemm(a, w, bias=bias, out16=o16))
        t2 = timeit(lambda: torch.matmul(a, w.t()))
        print(f"gemm {name:10s} {m}x{n}x{k}: {t*1e6:8.1f} us  {2*m*n*k/t/1e12:7.1f} TF/s   (torch/hipBLASLt {t2*1e6:8.1f} us {2*m*n*k/t2/1e12:7.1f} TF/s)")
    for (b, heads, g, name) in [(2, 12, 64, "global 2 img"), (8, 12, 64, "global 8 img")]:
        t_ = g * g
        e = heads * 64
        qkv = torch.randn(b * t_, 3 * e, device="cuda").to(dt)
        vt = torch.randn(b * heads, 64, t_, device="cuda").to(dt)
        out = torch.empty(b * t_, e, device="cuda", dtype=dt)
        relh = torch.randn(b * heads, t_, g, device="cuda")
        relw = torch.randn(b * heads, t_, g, device="cuda")
        tab = torch.randn(2 * g - 1, 64, device="cuda").to(dt)
        fl = 4 * b * heads * t_ * t_ * 64
        t = timeit(lambda: L.attn_fwd(qkv, vt, out, relh, relw, b, heads, t_, t_, g, e, 0.125, L.ATTN_RELPOS))
        print(f"attn relpos {name}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, t_, g, e, 0.125, L.ATTN_RELPOS, tabh=tab, tabw=tab))
        print(f"attn relpos in-kernel {name}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, t_, 0, e, 0.125, L.ATTN_PLAIN))
        print(f"attn plain  {name}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: L.relpos_terms(qkv, b, heads, g, e, tab, tab, relh, relw))
        print(f"relpos terms {name}: {t*1e6:8.1f} us")
    # windows
    b, heads, g = 50, 12, 14
    t_ = 196
    e = heads * 64
    qkv = torch.randn(b * t_, 3 * e, device="cuda").to(dt)
    vt = torch.randn(b * heads, 64, 256, device="cuda").to(dt)
    out = torch.empty(b * t_, e, device="cuda", dtype=dt)
    relh = torch.randn(b * heads, t_, g, device="cuda")
    relw = torch.randn(b * heads, t_, g, device="cuda")
    tab = torch.randn(2 * g - 1, 64, device="cuda").to(dt)
    fl = 4 * b * heads * t_ * t_ * 64
    t = timeit(lambda: L.attn_fwd(qkv, vt, out, relh, relw, b, heads, t_, 256, g, e, 0.125, L.ATTN_RELPOS))
    print(f"attn window (50 win): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, 256, g, e, 0.125, L.ATTN_RELPOS, tabh=tab, tabw=tab))
    print(f"attn window in-kernel bias (50 win): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: L.attn_fwd(qkv, vt, out, None, None, b, heads, t_, 256, g, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tab, tabw=tab))
    print(f"attn window slot-order WIN16 (50 win): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: L.relpos_terms(qkv, b, heads, g, e, tab, tab, relh, relw))
    print(f"relpos terms window: {t*1e6:8.1f} us")
    x = torch.randn(8192, 768, device="cuda")
    gm, bt = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
    o16 = torch.empty(8192, 768, device="cuda", dtype=dt)
    t = timeit(lambda: L.layernorm(x, gm, bt, 1e-6, out16=o16, dt=L._DT[dt]))
    print(f"layernorm 8192x768: {t*1e6:8.1f} us  {(8192*768*6)/t/1e9:7.1f} GB/s")
    x = torch.randn(65536, 768, device="cuda")
    o16 = torch.empty(65536, 768, device="cuda", dtype=dt)
    t = timeit(lambda: L.layernorm(x, gm, bt, 1e-6, out16=o16, dt=L._DT[dt]))
    print(f"layernorm 65536x768: {t*1e6:8.1f} us  {(65536*768*6)/t/1e9:7.1f} GB/s")
    # metrics: 8 x 1024 x 1024 label maps (blocky, like real masks, and random = worst case for the histogram atomics)
    from labelanything_amd.metrics import SegmentationMeter
    k = 81
    for name, gt in (("blocky", torch.randint(0, k, (8, 64, 64)).repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous().cuda()),
                     ("random", torch.randint(0, k, (8, 1024, 1024)).cuda())):
        pred = torch.roll(gt, 5, 2).contiguous()
        m = SegmentationMeter(k)
        t = timeit(lambda: m.update(pred, gt))
        print(f"confmat 8x1024x1024 K=81 {name}: {t*1e6:8.1f} us  {gt.numel()*16/t/1e9:7.1f} GB/s")
    # image preprocessing: COCO-size uint8 image -> 1024-long-side resize (PIL-exact) -> normalise -> pad, vs PIL + torch on the host
    import numpy as np
    import time
    from PIL import Image
    from labelanything_amd.image_prep import DevicePreprocessor
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    dp = DevicePreprocessor(1024, True, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], square=False)
    dimg = torch.from_numpy(img).cuda()
    t = timeit(lambda: dp(dimg))
    t0 = time.perf_counter()
    for _ in range(10):
        x = torch.from_numpy(np.asarray(Image.fromarray(img).resize((1024, 768), Image.BILINEAR)).copy()).permute(2, 0, 1).float() / 255.0
        x = torch.nn.functional.pad((x - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1), (0, 0, 0, 256))
    tc = (time.perf_counter() - t0) / 10
    print(f"image prep 480x640 -> 3x1024x1024: device {t*1e6:8.1f} us ({(480*640*3 + 3*1024*1024*4)/t/1e9:6.1f} GB/s)   PIL + torch on one host core {tc*1e3:6.1f} ms")


if __name__ == "__main__":
    main()
