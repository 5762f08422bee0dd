#!/usr/bin/env python
"""Achieved HBM bandwidth of the streaming kernels (LayerNorm, GroupNorm passes, concat) at the UNet
level-0 / level-1 sizes of the E=10 benchmark, against a plain device-to-device copy.  Tuning tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (M, C) in ((92160, 320), (23040, 640), (5760, 1280), (92160, 640)):
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    nbytes = M * C * 2
    t = timeit(lambda: y.copy_(x))
    print(f"M={M} C={C}: copy {2 * nbytes / t / 1e9:7.0f} GB/s", end="  ")
    op = O.layernorm(x, g, b, y, M=M, C=C)
    t = timeit(lambda: O.launch(op))
    print(f"layernorm {2 * nbytes / t / 1e9:7.0f} GB/s ({t * 1e6:.1f} us)", end="  ")
    B, HW = 10, M // 10
    ss = torch.empty(B, 2, C, device=dev)
    for target in (256, 384, 512, 768):
        chunks = max(1, min(HW // 32, max(16, target // B)))
        part = torch.empty(B, chunks, C, 2, device=dev)
        op = O.gn_stats(x, part, B=B, HW=HW, C=C, chunks=chunks)
        t = timeit(lambda: O.launch(op))
        op2 = O.gn_finalize(part, g, b, ss, B=B, C=C, groups=32, chunks=chunks, HW=HW, eps=1e-5)
        t2 = timeit(lambda: O.launch(op2))
        print(f"gn_stats[{target}] {nbytes / t / 1e9:5.0f} GB/s ({t * 1e6:.1f}+{t2 * 1e6:.1f} us)", end="  ")
    op = O.gn_apply(x, ss, y, B=B, HW=HW, C=C, silu=True)
    t = timeit(lambda: O.launch(op))
    print(f"gn_apply {2 * nbytes / t / 1e9:7.0f} GB/s ({t * 1e6:.1f} us)")
