#!/usr/bin/env python
"""Adjoint of the dense mask embedding's 64 x 64 -> 30 x 30 resize at the cfg3 training shape (76800 planes): scatter with atomics into a
zero-filled destination (la_bilinear_bwd) against the gather (la_bilinear_bwd_set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd import _lib as L
n, ih, iw, oh, ow = 76800, 64, 64, 30, 30
dy = torch.randn(n, oh, ow, device="cuda")


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def scatter():
    dx = dy.new_zeros(n, ih, iw)
    L.bilinear_bwd(dy, n, oh, ow, oh * ow, ow, dx, ih, iw, ih * iw, iw)
    return dx


def gather():
    dx = dy.new_empty(n, ih, iw)
    L.bilinear_bwd_set(dy, n, oh, ow, oh * ow, ow, dx, ih, iw, ih * iw, iw)
    return dx


a, b = scatter(), gather()
print(f"zero fill + scatter {timed(scatter):8.1f} us | gather {timed(gather):8.1f} us | max difference {float((a - b).abs().max()):.2e} of {float(a.abs().max()):.2e}")
