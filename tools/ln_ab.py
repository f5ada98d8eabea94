import os, sys
sys.path.insert(0, os.getcwd())
import torch
from labelanything_amd import _lib as L
def bench(fn, it=20):
    for _ in range(3): fn()
    ts=[]
    for _ in range(5):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)/it*1e3)
    return sorted(ts)[2]
for rows,e,rpg in ((262144,768,4096),(57664,768,901),(46852,1024,901)):
    x=torch.randn(rows,e,device="cuda"); g=torch.ones(e,device="cuda"); b=torch.zeros(e,device="cuda")
    rv=torch.randn(rows//rpg,e,device="cuda")*0.01
    o16=torch.empty(rows,e,device="cuda",dtype=torch.float16)
    part=torch.empty(rows//rpg*L.ln_cs_chunks(rpg)*e,device="cuda")
    t0=bench(lambda: L.layernorm(x,g,b,1e-6,out16=o16))
    t1=bench(lambda: L.layernorm_g(x,rv,rpg,g,b,1e-6,out16=o16))
    t2=bench(lambda: L.layernorm_g(x,rv,rpg,g,b,1e-6,out16=o16,colsum_part=part))
    gb=rows*e*6/1e3
    print(f"LN {rows}x{e}: plain {t0:7.1f} us ({gb/t0:5.0f} GB/s)  +rvec {t1:7.1f}  +colsum {t2:7.1f}  checksum {float(o16.float().abs().sum()):.6e}")
