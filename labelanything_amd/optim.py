"""Optimizer side of training: the reference trains with ``torch.optim.AdamW`` (defaults, lr 5e-5) and
HF's ``constant_with_warmup`` schedule (``experiment/utils.py:53-100``, ``parameters/trainval/coco20i/mae_noembs.yaml:32-37``)
under DDP, i.e. one gradient all-reduce per optimizer step.  Here the learnable parameters live in ONE flat fp32 buffer (the
model's tensors become views of it), so a step is: one SUM all-reduce of the flat gradient over RCCL (``sum_over_ranks``) + one
``la_adamw_step`` launch that also applies the 1 / world scaling.  The gradient is written by ``train.LamTrainer``'s backward pass
(autograd accumulates straight into ``grad_views``).
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _lib as L
from .parallel import sum_over_ranks


def constant_with_warmup(step: int, num_warmup_steps: int) -> float:
    """lr multiplier of transformers' ``get_constant_schedule_with_warmup`` after ``step`` scheduler steps."""
    if step < num_warmup_steps:
        return float(step) / float(max(1.0, num_warmup_steps))
    return 1.0


class FlatAdamW:
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 5e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 num_warmup_steps: int = 0, lrs=None):
        """lrs: optional per-tensor base learning rates (the reference's ``backbone_lr`` parameter group, models/lam.py:340-346); the
        warm-up factor multiplies every group's rate alike, as the HF scheduler does."""
        self.params: List[torch.Tensor] = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda" or any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise RuntimeError("FlatAdamW needs fp32 device parameters (there is no CPU path)")
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        off = 0
        self.views, self.grad_views = [], []
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            view = self.flat[off:off + k].view_as(p)
            p.data = view                                    # the model now reads / the optimizer writes the same storage
            self.views.append(view)
            self.grad_views.append(self.grad[off:off + k].view_as(p))
            off += k
        self.base_lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.lrs = [float(lr)] * len(self.params) if lrs is None else [float(v) for v in lrs]
        if len(self.lrs) != len(self.params):
            raise ValueError("lrs needs one rate per parameter tensor")
        self.num_warmup_steps = int(num_warmup_steps)
        self.keep_reduced_grad = False  # tests: keep a copy of the all-reduced, world-averaged gradient of the last step
        self.reduced_grad = None
        self.steps = 0                 # optimizer steps taken
        self.tensor_steps = [0] * len(self.params)   # torch.optim.AdamW keeps state['step'] PER parameter: it only advances on
        #                                              steps where the tensor has a gradient, and drives that tensor's bias correction
        self.sched_steps = 0           # scheduler steps taken (the reference steps it per batch: mae_noembs.yaml:36)

    @property
    def lr(self) -> float:
        return self.base_lr * constant_with_warmup(self.sched_steps, self.num_warmup_steps) if self.num_warmup_steps else self.base_lr

    def zero_grad(self) -> None:
        self.grad.zero_()

    def step(self, all_reduce: bool = True, active=None, reducer=None) -> None:
        """Average ``self.grad`` over the data-parallel ranks (if a process group is up) and apply one AdamW update.

        active: optional per-parameter flags "this tensor received a gradient in this step".  torch.optim.AdamW skips tensors whose
        ``.grad`` is None - no moment update and NO weight decay - which is what happens to the parameters the reference's forward
        never reaches (DDP ``find_unused_parameters``, experiment/run.py:123); contiguous runs of active tensors are updated with
        one launch each (one launch in the usual case: never-used tensors are kept at the tail by LamTrainer).

        reducer: a ``parallel.BucketedGradReducer`` over ``self.grad`` - buckets that were launched during / right after the backward
        pass are waited for one at a time, each followed by the AdamW launches of ITS tensors, so the collective of the next bucket
        overlaps the optimizer kernel of this one; without it the whole buffer is reduced by one collective here."""
        world = 1
        if all_reduce and torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
            if reducer is None:
                sum_over_ranks(self.grad)
        if reducer is None or not all_reduce:
            if reducer is not None:              # collectives already launched by a synchronising micro-step still write into self.grad
                reducer.finish_all()
            bucket_bounds = [(0, self.grad.numel())]
            reducer = None
        else:
            bucket_bounds = reducer.bounds
        if self.keep_reduced_grad:
            if reducer is not None:
                reducer.finish_all()
            self.reduced_grad = self.grad / float(world)
        self.steps += 1
        if active is None:
            active = [True] * len(self.params)
        # contiguous runs of active tensors that share a step count (and a learning rate) -> one launch each
        spans, off, cur = [], 0, None
        for i, (p, a) in enumerate(zip(self.params, active)):
            k = p.numel()
            if a:
                self.tensor_steps[i] += 1
                t = self.tensor_steps[i]
                if cur is not None and cur[2] == t and cur[3] == self.lrs[i]:
                    cur[1] = off + k
                else:
                    if cur is not None:
                        spans.append(tuple(cur))
                    cur = [off, off + k, t, self.lrs[i]]
            elif cur is not None:
                spans.append(tuple(cur))
                cur = None
            off += k
        if cur is not None:
            spans.append(tuple(cur))
        factor = constant_with_warmup(self.sched_steps, self.num_warmup_steps) if self.num_warmup_steps else 1.0
        for bi, (ba, bb) in enumerate(bucket_bounds):
            if reducer is not None:
                reducer.finish(bi)                   # this stream waits for bucket bi only; later buckets keep travelling
            for a, b, t, lr in spans:
                a, b = max(a, ba), min(b, bb)        # (the update is element-wise: a span cut at a bucket boundary is the same update)
                if a < b:
                    L.adamw_step(self.flat[a:b], self.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], lr * factor, self.betas[0],
                                 self.betas[1], self.eps, self.weight_decay, t, 1.0 / world)
        self.sched_steps += 1
