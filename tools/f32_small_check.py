import sys, torch
sys.path.insert(0, "/root/repo")
from labelanything_amd import _lib as L
torch.manual_seed(0)
for (m, n, k) in ((52, 768, 1536), (26, 768, 1536), (52, 768, 768), (300, 256, 256), (150, 2048, 256), (300, 256, 2048)):
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5
    r = torch.randn(m, n, device="cuda")
    out = r.clone()
    L.gemm(a, w, res=out, out32=out)
    ref = r.double() + a.double() @ w.double().t()
    print(m, n, k, float((out.double() - ref).abs().max() / ref.abs().max()))
