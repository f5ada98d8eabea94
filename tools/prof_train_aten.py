import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from torch.profiler import profile, ProfilerActivity
from labelanything_amd.train import LamTrainer
enc = "--train-encoder" in sys.argv
wl = next((a.split("=")[1] for a in sys.argv if a.startswith("--workload=")), "cfg3_train")
lam, cfg = bench.build_model(torch.float16, torch.float32, wl, None)
lam = lam.cuda()
batch = bench.make_inputs(2, 1234, torch.device("cuda"), wl)
tr = LamTrainer(lam, lr=5e-5, num_warmup_steps=1000, train_encoder=enc)
size = bench.WORKLOADS[wl]["episode"]["image_size"]
gt = torch.randint(0, batch["flag_examples"].shape[2], (2, size, size)).cuda()
for _ in range(2):
    tr.step(batch, gt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        tr.step(batch, gt)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6) if e.device_time_total > 0 and e.key.startswith("aten::")]
ev.sort(key=lambda e: -e.device_time_total)
for e in ev[:28]:
    st = [s for s in e.stack if "labelanything_amd" in s or "bench" in s][:2]
    print(f"{e.device_time_total / 2e3:8.3f} ms/step  x{e.count // 2:4d}  {e.key:28s} {str(e.input_shapes)[:70]:70s} {' | '.join(s.split('/')[-1][:60] for s in st)}")
