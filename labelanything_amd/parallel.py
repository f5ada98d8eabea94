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


class BucketedGradReducer:
    """SUM all-reduce of one flat gradient buffer in BUCKETS (contiguous element ranges) that are launched as soon as their gradients
    are final and waited for one by one, so that the collective of bucket i + 1 runs while the optimizer kernel of bucket i does
    (SURVEY 8e "overlapped with the tail of backward"; the reference gets the same from DDP's gradient buckets,
    experiment/run.py:122-131,172).

    * RCCL ("nccl" backend): ``launch`` enqueues the collective from a side stream that first waits for the caller's stream, and
      returns at once; ``finish`` makes the caller's stream wait for it.  No host synchronisation anywhere.
    * ``staged=True`` reduces a COPY of the bucket: used for the bucket launched from INSIDE the backward pass (the decoder-side
      gradients, final before the encoder backward starts).  If a gradient of that bucket is written after the launch
      (``invalidate``), ``finish`` drops the copy and reduces the bucket in place instead - the result is always the sum of the final
      gradients.
    * gloo (CPU tests, two processes on one GPU) has no device path: buckets are reduced synchronously in ``finish``.
    Every element is reduced exactly once per ``begin`` / ``finish_all`` cycle; with two ranks the result is bit-identical to one
    collective over the whole buffer (a + b either way)."""

    def __init__(self, grad: torch.Tensor, bounds: Sequence[Sequence[int]], single_rank_collectives: bool = False):
        """single_rank_collectives: issue the collectives also in a process group of ONE rank (they are identities there) - lets a
        one-GPU box run the RCCL side-stream choreography (tests)."""
        self.single_rank_collectives = bool(single_rank_collectives)
        if grad.dim() != 1:
            raise ValueError("flat 1-D gradient buffer expected")
        self.grad = grad
        self.bounds = [(int(a), int(b)) for a, b in bounds]
        pos = 0
        for a, b in self.bounds:
            if a != pos or b <= a:
                raise ValueError("buckets must tile the buffer in order")
            pos = b
        if pos != grad.numel():
            raise ValueError("buckets must cover the whole buffer")
        self.side = torch.cuda.Stream(grad.device) if grad.is_cuda else None
        self._work = {}          # bucket -> (work handle | None, staging tensor | None)
        self._stale = set()      # staged buckets whose gradients changed after the launch
        self._done = set()

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.single_rank_collectives)

    def _async_ok(self) -> bool:
        return self.active() and not (self.grad.is_cuda and dist.get_backend() == "gloo")

    def begin(self) -> None:
        """Start of a reduction cycle (after ``zero_grad``): nothing launched, nothing reduced."""
        self._work.clear()
        self._stale.clear()
        self._done.clear()

    def launched(self, i: int) -> bool:
        return i in self._work or i in self._done

    def launch(self, i: int, staged: bool = False) -> None:
        """The gradients of bucket i are final (staged: probably final): start their all-reduce."""
        if self.launched(i):
            raise RuntimeError(f"bucket {i} was already reduced in this cycle (a second micro-step after a synchronising one?)")
        if not self._async_ok():
            return                                    # reduced synchronously in finish()
        a, b = self.bounds[i]
        buf = self.grad[a:b]
        if staged:
            buf = buf.clone()
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.grad.device))
            with torch.cuda.stream(self.side):
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self._work[i] = (work, buf if staged else None)

    def invalidate(self, i: int) -> None:
        """A gradient inside staged bucket i was written after its launch."""
        if i in self._work and self._work[i][1] is not None:
            self._stale.add(i)

    def stale_flags(self) -> List[bool]:
        """Per bucket: its staged copy was invalidated on THIS rank (exchanged over the ranks by the caller, then ``set_stale``)."""
        return [i in self._stale for i in range(len(self.bounds))]

    def set_stale(self, flags: Sequence[bool]) -> None:
        """The collective decision: a staged bucket that went stale on ANY rank is re-reduced in place on EVERY rank."""
        for i, f in enumerate(flags):
            if f and i in self._work and self._work[i][1] is not None:
                self._stale.add(i)

    def finish(self, i: int) -> None:
        """Bucket i holds the sum over the ranks when the caller's stream reaches this point."""
        if i in self._done:
            return
        a, b = self.bounds[i]
        if i in self._work:
            work, staging = self._work.pop(i)
            work.wait()
            if staging is None:
                self._done.add(i)
                return
            if i not in self._stale:
                self.grad[a:b].copy_(staging)
                self._done.add(i)
                return
        if self.active():
            if dist.get_world_size() > 1:
                sum_over_ranks(self.grad[a:b])
            else:
                dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM)
        self._done.add(i)

    def finish_all(self) -> None:
        for i in range(len(self.bounds)):
            self.finish(i)
