"""Episode sharding over the GPUs of one node.

Inference is embarrassingly parallel over episodes: one process per GPU, every rank takes a disjoint slice of the
episode list, no collective on the data path.  The only communication is bookkeeping (a barrier and a MAX / SUM
all-reduce of scalars), which runs over RCCL ("nccl" backend) on MI355X and over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def shard_episodes(n_episodes: int, rank: int, world: int) -> List[int]:
    """Round-robin episode indices of this rank (balanced to within one episode)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_episodes, world))


def slice_batch(batch: Dict[str, torch.Tensor], idx: Sequence[int]) -> Dict[str, torch.Tensor]:
    """Select episodes ``idx`` (batch dimension 0) from a reference-schema batch dict."""
    sel = torch.as_tensor(list(idx), dtype=torch.long)
    out = {}
    for k, v in batch.items():
        out[k] = v.index_select(0, sel.to(v.device)) if isinstance(v, torch.Tensor) else [v[i] for i in idx]
    return out


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (the bench's step time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _all_reduce(t: torch.Tensor, op) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() == "gloo":
            # gloo (CPU tests, and the two-processes-on-one-GPU tests) moves host memory: stage device tensors through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)
    return t


def sum_over_ranks(t: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce, in place (the flat gradient; a confusion matrix accumulated per rank, utils/metrics.py:28-51 in the reference)."""
    return _all_reduce(t, dist.ReduceOp.SUM)


def any_over_ranks(flags: Sequence[bool], device=None) -> List[bool]:
    """Element-wise OR of per-rank boolean flags (MAX all-reduce of a byte vector): what DDP's ``find_unused_parameters=True``
    does with its used-parameter bitmap (experiment/run.py:123) - a parameter counts as used if ANY rank used it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [bool(f) for f in flags]
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    return [bool(v) for v in t.cpu().tolist()]
