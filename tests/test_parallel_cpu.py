"""CPU, world_size 2 over gloo: the episode sharding used for N > 1 GPUs (no data-path collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from labelanything_amd.episodes import make_episode
from labelanything_amd.parallel import shard_episodes, slice_batch, max_over_ranks, sum_over_ranks


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_episodes: int, out_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = make_episode(batch=n_episodes, n_ways=1, k_shots=1, image_size=32, seed=5, prompts=("mask",))
    mine = shard_episodes(n_episodes, rank, world)
    sub = slice_batch(batch, mine)
    assert sub["images"].shape[0] == len(mine)
    assert torch.equal(sub["images"], batch["images"][mine])
    # bookkeeping collectives: every episode is processed exactly once across the job
    seen = torch.zeros(n_episodes, dtype=torch.int64)
    seen[mine] = 1
    seen = sum_over_ranks(seen)
    assert bool((seen == 1).all())
    t = max_over_ranks(1.0 + rank)
    assert t == float(world)
    dist.barrier()
    torch.save({"rank": rank, "mine": mine}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_episode_sharding(tmp_path):
    world, n = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    got = sorted(i for r in range(world) for i in torch.load(tmp_path / f"r{r}.pt")["mine"])
    assert got == list(range(n))


def test_shard_is_balanced_and_disjoint():
    for n in (1, 7, 8, 33):
        for world in (1, 2, 4, 8):
            parts = [shard_episodes(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _meter_worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist
    from labelanything_amd.metrics import metrics_from_state
    from labelanything_amd.parallel import sum_over_ranks
    from oracle import metrics_oracle as MO
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    p, g = rng.integers(0, 4, 500), rng.integers(0, 4, 500)
    cm = torch.from_numpy(MO.confusion_matrix(p, g, 4)).clone()
    cb = torch.from_numpy(MO.binary_confusion_matrix(p, g)).clone()
    sum_over_ranks(cm)
    sum_over_ranks(cb)
    q.put((rank, cm.numpy().tolist(), metrics_from_state(cm, cb)))
    dist.destroy_process_group()


def test_confusion_matrix_all_reduce_two_ranks():
    """the only collective of the inference path - a SUM all-reduce of the (K+1)^2 int64 confusion matrix."""
    import numpy as np
    import torch.multiprocessing as mp
    from oracle import metrics_oracle as MO
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_meter_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    expect = 0
    for r in range(2):
        rng = np.random.default_rng(100 + r)
        pp, gg = rng.integers(0, 4, 500), rng.integers(0, 4, 500)
        expect = expect + MO.confusion_matrix(pp, gg, 4)
    assert res[0][1] == res[1][1] == expect.tolist()
    assert abs(res[0][2]["mIoU"] - MO.strict_mean_iou(expect)) < 1e-6


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    from labelanything_amd.parallel import sum_over_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7 + rank)
    flat_grad = torch.randn(1000, generator=g)            # what FlatAdamW.grad holds after a backward pass
    sum_over_ranks(flat_grad)
    q.put((rank, (flat_grad / world).tolist()))
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_two_ranks():
    """SURVEY 8e, training: ONE all-reduce (sum, then / world) of the flat gradient buffer per optimizer step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    expect = sum(torch.randn(1000, generator=torch.Generator().manual_seed(7 + r)) for r in range(2)) / 2
    assert res[0][1] == res[1][1]
    assert float((torch.tensor(res[0][1]) - expect).abs().max()) < 1e-6


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    from labelanything_amd.parallel import BucketedGradReducer, sum_over_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(70 + rank)
    base = torch.randn(10_000, generator=g)
    single = base.clone()
    sum_over_ranks(single)                                   # the one-collective form of round 3
    # (a) four in-place buckets + one STAGED bucket launched early, all waited in order
    grad = base.clone()
    red = BucketedGradReducer(grad, [(0, 2000), (2000, 4100), (4100, 6000), (6000, 8192), (8192, 10_000)])
    red.begin()
    red.launch(4, staged=True)                               # "decoder" bucket from inside the backward pass
    for i in range(4):
        red.launch(i)                                        # "encoder" buckets right behind it
    for i in range(5):
        red.finish(i)
    same_a = torch.equal(grad, single)
    # (b) a gradient written into the staged bucket AFTER its launch: the copy is dropped, the bucket is reduced in place
    grad2 = base.clone()
    red2 = BucketedGradReducer(grad2, [(0, 8192), (8192, 10_000)])
    red2.begin()
    red2.launch(1, staged=True)
    grad2[9000:9100] += 1.0 + rank                           # late accumulation (both ranks, different values)
    red2.invalidate(1)
    red2.launch(0)
    red2.finish_all()
    expect2 = base.clone()
    expect2[9000:9100] += 1.0 + rank
    sum_over_ranks(expect2)
    same_b = torch.equal(grad2, expect2)
    # (c) nothing launched before the optimizer step: finish_all reduces everything; a double launch is refused
    grad3 = base.clone()
    red3 = BucketedGradReducer(grad3, [(0, 5000), (5000, 10_000)])
    red3.begin()
    red3.finish_all()
    same_c = torch.equal(grad3, single)
    red3.begin()
    red3.launch(0)
    try:
        red3.launch(0)
        refused = False
    except RuntimeError:
        refused = True
    red3.finish_all()
    # (d) ADVICE r4: the staged bucket goes stale on rank 0 ONLY - the decision is exchanged (as LamTrainer.apply_update does, in the
    # used-parameter exchange) and BOTH ranks re-reduce in place: same collective sequence on every rank, sum of the final gradients
    from labelanything_amd.parallel import any_over_ranks
    grad4 = base.clone()
    red4 = BucketedGradReducer(grad4, [(0, 8192), (8192, 10_000)])
    red4.begin()
    red4.launch(1, staged=True)
    if rank == 0:
        grad4[9000:9100] += 2.5
        red4.invalidate(1)
    red4.launch(0)
    local = red4.stale_flags()
    red4.set_stale(any_over_ranks(local))
    red4.finish_all()
    expect4 = base.clone()
    if rank == 0:
        expect4[9000:9100] += 2.5
    sum_over_ranks(expect4)
    same_d = torch.equal(grad4, expect4) and local == [False, rank == 0] and red4.stale_flags() == [False, True]
    q.put((rank, same_a, same_b, same_c and same_d, refused))
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_equals_the_single_collective():
    """the flat gradient reduced in buckets (one of them staged and launched early, as LamTrainer does for the
    decoder-side gradients) is bit-identical to the single collective on two ranks, also when a late gradient invalidates the staged
    copy."""
    import pytest
    import torch.multiprocessing as mp
    from labelanything_amd.parallel import BucketedGradReducer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True, True, True), (1, True, True, True, True)]
    with pytest.raises(ValueError):
        BucketedGradReducer(torch.zeros(10), [(0, 4), (5, 10)])        # buckets must tile the buffer
    one = BucketedGradReducer(torch.ones(10), [(0, 10)])               # no process group: everything is a no-op
    one.begin()
    one.launch(0, staged=True)
    one.finish_all()
    assert torch.equal(one.grad, torch.ones(10))
