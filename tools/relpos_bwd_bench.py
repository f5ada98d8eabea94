import torch, sys
sys.path.insert(0, "/root/repo")
from labelanything_amd import _lib as L
for b, heads, g in ((4, 12, 64), (100, 12, 14)):
    t, e = g * g, heads * 64
    qkv = torch.randn(b * t, 3 * e, device="cuda").half()
    dqkv = torch.zeros_like(qkv)
    drh = torch.randn(b * heads, t, g, device="cuda"); drw = torch.randn_like(drh)
    th = torch.randn(2 * g - 1, 64, device="cuda"); tw = torch.randn_like(th)
    dth = torch.zeros_like(th); dtw = torch.zeros_like(tw)
    f = lambda: L.relpos_bwd(qkv, dqkv, drh, drw, th, tw, dth, dtw, b, heads, g, e)
    for _ in range(3): f()
    s, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e2.record(); torch.cuda.synchronize()
    print(f"relpos_bwd B={b} heads={heads} G={g}: {s.elapsed_time(e2) / 10 * 1e3:.1f} us")
