"""GPU, two processes on the ONE GPU of the box (gloo rendezvous): the data-parallel trainer and the bench's multi-rank path.

Reference semantics (experiment/run.py:122-131,359-361): DDP with ``find_unused_parameters=True`` averages the gradients over the
ranks and treats a parameter as used when ANY rank used it; prompt types are sampled per episode (data/dataset.py:292), so the ranks
of one step routinely disagree about which prompt-encoder tensors their forward touched."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from labelanything_amd.episodes import make_episode
from tests.cases import TRAIN_CASE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 2


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _episode(rank: int, step: int):
    """Rank 0 sees masks + points + boxes, rank 1 masks only: point_embeddings.*, not_a_point_embed are touched on rank 0 alone,
    no_sparse_embedding on rank 1 alone."""
    ep = dict(TRAIN_CASE["episode"])
    ep.update(batch=1, seed=500 + 10 * step + rank, prompts=("mask", "point", "box") if rank == 0 else ("mask",))
    batch = make_episode(**ep)
    c = batch["flag_examples"].shape[2]
    g = torch.Generator().manual_seed(900 + 10 * step + rank)
    h, w = int(batch["dims"][0, 0, 0]), int(batch["dims"][0, 0, 1])
    gt = torch.randint(0, c, (1, h // 4, w // 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2)
    return batch, gt


def _trainer():
    from labelanything_amd.models import Lam
    from labelanything_amd.train import LamTrainer
    lam = Lam(TRAIN_CASE["cfg"], seed=TRAIN_CASE["weight_seed"]).cuda()
    lam.selected_rows = torch.tensor([1, 4, 7])
    return lam, LamTrainer(lam, lr=1e-3, weight_decay=1e-2, num_warmup_steps=2)


def _worker(rank: int, world: int, port: int, out_dir: str):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lam, tr = _trainer()
    tr.opt.keep_reduced_grad = True
    grads = []
    for step in range(STEPS):
        batch, gt = _episode(rank, step)
        tr.step(batch, gt)
        grads.append(tr.opt.reduced_grad.cpu())
    torch.cuda.synchronize()
    torch.save({"flat": tr.opt.flat.cpu(), "steps": list(tr.opt.tensor_steps), "names": tr.names, "grad0": grads[0]},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_keeps_replicas_identical_with_rank_local_prompt_types(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in range(world))
    # (a) replicas bit-identical, including the tensors only one of the ranks touched
    assert torch.equal(r0["flat"], r1["flat"])
    assert r0["steps"] == r1["steps"]
    names = r0["names"]
    by = dict(zip(names, r0["steps"]))
    assert by["prompt_encoder.point_embeddings.0.weight"] == STEPS and by["prompt_encoder.no_sparse_embedding.weight"] == STEPS
    assert by["prompt_encoder.transformer.norm_final_attn.weight"] == 0          # dead on every rank: never stepped, never decayed
    # (b) == one process that runs the two ranks' episodes as two accumulated micro-steps (mean of the per-rank gradients, flags
    # OR-ed over the micro-steps) and then updates once.  NOT the concatenated 2-episode batch: that is a different function in
    # the reference as well (the padded "not a point" tokens of the episode without points, the per-batch class weighting of the
    # objective) - DDP is the average of per-rank objectives
    lam, tr = _trainer()
    tr.opt.keep_reduced_grad = True
    for step in range(STEPS):
        tr.zero_grad()
        for rank in range(world):
            batch, gt = _episode(rank, step)
            tr.forward_backward(batch, gt, loss_normalizer=float(world))
        tr.apply_update()
        if step == 0:
            g_single = tr.opt.reduced_grad.cpu()
    torch.cuda.synchronize()
    single = tr.opt.flat.cpu()
    assert tr.opt.tensor_steps == r0["steps"]
    assert torch.equal(r0["grad0"], r1["grad0"])
    gs = float(g_single.abs().max())
    # (the same fp32 sums in a different grouping: two half-batches added by the all-reduce vs one batch through the atomics of the
    # weight / LayerNorm gradient kernels - measured 1.6 - 2.1e-6 of the largest entry)
    assert float((g_single - r0["grad0"]).abs().max()) <= 5e-6 * gs, float((g_single - r0["grad0"]).abs().max()) / gs
    # parameters after two AdamW steps: lr * m / (sqrt(v) + eps) turns a rounding-level difference of a near-zero gradient entry into
    # a difference of up to ~lr per step; every entry stays within that, and all but a sliver agree to rounding
    diff = (single - r0["flat"]).abs()
    assert float(diff.max()) <= 2 * STEPS * 1e-3 * 0.1, float(diff.max())
    assert float((diff > 1e-6 * float(single.abs().max())).float().mean()) <= 0.02


def test_bench_runs_as_two_ranks_and_prints_one_line(tmp_path):
    """The driver's N > 1 launch (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N) with both ranks on this
    box's one GPU and gloo instead of RCCL: exactly one JSON line, n_gpus 2, whole-job value = both ranks' episodes."""
    env = dict(os.environ, LA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--episodes", "1", "--workload", "cfg1", "--no-cpu-baseline", "--no-eager-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["global_episodes_per_step"] == 2
    assert abs(rec["value"] - 2 / (rec["ms_per_step"] * 1e-3)) <= 1e-2 * rec["value"]


def test_bench_launches_its_own_ranks(tmp_path):
    """Plain `python bench.py --gpus 2` (no torch.distributed.run around it, WORLD_SIZE unset): bench.py starts the two ranks itself,
    refuses more ranks than GPUs unless LA_BENCH_BACKEND=gloo, and refuses a WORLD_SIZE that contradicts --gpus."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    args = ["--steps", "2", "--warmup", "1", "--episodes", "1", "--workload", "cfg1", "--no-cpu-baseline", "--no-eager-baseline"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_episodes_per_step"] == 2
    import torch
    if torch.cuda.device_count() < 2:
        env_nccl = dict(env)
        env_nccl.pop("LA_BENCH_BACKEND")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, cwd=ROOT, env=env_nccl,
                             capture_output=True, text=True, timeout=300)
        assert res.returncode != 0 and "GPU(s) visible" in (res.stderr + res.stdout)
    env_bad = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, cwd=ROOT, env=env_bad, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


# ---- bucketed gradient all-reduce ---------------------------------------------------------------------------------
def _enc_trainer(buckets: int):
    from labelanything_amd.models import Lam
    from labelanything_amd.train import LamTrainer
    from tests.cases import TRAIN_ENC_CASE as case
    lam = Lam(case["cfg"], seed=case["weight_seed"]).cuda()
    return lam, LamTrainer(lam, lr=1e-3, weight_decay=1e-2, num_warmup_steps=2, train_encoder=True, encoder_buckets=buckets)


def _enc_episode(rank: int, step: int):
    from tests.cases import TRAIN_ENC_CASE as case
    from tests.test_train_gpu import make_gt
    ep = dict(case["episode"])
    ep.update(seed=700 + 10 * step + rank, prompts=("mask", "point") if rank == 0 else ("mask",))
    batch = make_episode(**ep)
    return batch, make_gt(batch, batch["flag_examples"].shape[2], seed=11 + rank + 2 * step)


def _enc_worker(rank: int, world: int, port: int, out_dir: str):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    out = {}
    for buckets in (4, 1):
        lam, tr = _enc_trainer(buckets)
        tr.opt.keep_reduced_grad = True
        g0 = None
        for step in range(STEPS):
            tr.step(*_enc_episode(rank, step))               # forward_backward(sync=True): staged decoder bucket + encoder buckets
            if step == 0:
                g0 = tr.opt.reduced_grad.cpu()
        torch.cuda.synchronize()
        out[buckets] = {"flat": tr.opt.flat.cpu(), "grad": tr.opt.reduced_grad.cpu(), "grad0": g0, "bounds": tr.reducer.bounds}
    torch.save(out, os.path.join(out_dir, f"enc{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_with_trainable_encoder_and_gradient_buckets(tmp_path):
    """Two ranks (two processes on this GPU), everything trainable: the flat gradient travels in 4 encoder buckets + the decoder-side
    bucket that is launched from inside the backward pass; replicas stay bit-identical, and the run equals the one-bucket run to the
    run-to-run noise of the backward pass itself (its weight / LayerNorm gradients are fp32 atomic sums: two runs of the SAME
    configuration differ in the last bits; the reducer alone is bit-identical to one collective - tests/test_parallel_cpu.py)."""
    import torch.multiprocessing as mp
    mp.spawn(_enc_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"enc{r}.pt") for r in range(2))
    assert len(r0[4]["bounds"]) == 5 and len(r0[1]["bounds"]) == 2
    assert r0[4]["bounds"][-1][1] == r0[4]["flat"].numel() and all(a[1] == b[0] for a, b in zip(r0[4]["bounds"], r0[4]["bounds"][1:]))
    for b in (4, 1):
        assert torch.equal(r0[b]["flat"], r1[b]["flat"]) and torch.equal(r0[b]["grad"], r1[b]["grad"])
    gs = float(r0[1]["grad0"].abs().max())
    assert float((r0[4]["grad0"] - r0[1]["grad0"]).abs().max()) <= 1e-5 * gs
    diff = (r0[4]["flat"] - r0[1]["flat"]).abs()                # AdamW turns a rounding-level gradient difference into <= ~lr per step
    assert float(diff.max()) <= 2 * STEPS * 1e-3 and float((diff > 1e-6 * float(r0[1]["flat"].abs().max())).float().mean()) <= 0.02


def _rccl_single_worker(rank: int, world: int, port: int, out_dir: str):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    flats = {}
    for mode in ("bucketed", "plain"):
        lam, tr = _enc_trainer(4)
        tr.reducer.single_rank_collectives = mode == "bucketed"     # world size 1: the collectives are identities, the streams are real
        for step in range(STEPS):
            tr.step(*_enc_episode(0, step))
        torch.cuda.synchronize()
        flats[mode] = tr.opt.flat.cpu()
    torch.save(flats, os.path.join(out_dir, "rccl1.pt"))
    dist.destroy_process_group()


def test_rccl_side_stream_choreography_on_one_rank(tmp_path):
    """The RCCL path of BucketedGradReducer (async collectives from a side stream, staged decoder bucket launched under the encoder
    backward, per-bucket waits in front of the AdamW launches) on a process group of ONE rank - all this box can host: the training
    result must equal the run without any collective (to the run-to-run noise of the backward's atomic sums, as above)."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_single_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    f = torch.load(tmp_path / "rccl1.pt")
    diff = (f["bucketed"] - f["plain"]).abs()
    assert bool(torch.isfinite(f["bucketed"]).all())
    assert float(diff.max()) <= 2 * STEPS * 1e-3 and float((diff > 1e-6 * float(f["plain"].abs().max())).float().mean()) <= 0.02
