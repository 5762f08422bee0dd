#!/usr/bin/env python
"""Pure-GEMM comparison on one box (run on the MI355X): n^3 bf16 GEMMs on chosen tile variants of this library's implicit
GEMM against torch.matmul (= hipBLASLt), interleaved round-robin, uniform random [-1, 1) operands.
  GEMM_VARIANTS=62,72 GEMM_SIZES=4096,8192 python tools/gemm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
VARIANTS = tuple(int(v) for v in os.environ.get("GEMM_VARIANTS", "0,62,72").split(","))
SIZES = tuple(int(v) for v in os.environ.get("GEMM_SIZES", "4096,8192").split(","))
ROUNDS = int(os.environ.get("GEMM_ROUNDS", "5"))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n in SIZES:
    g = torch.Generator(device="cpu").manual_seed(7)
    a = (torch.rand(n, n, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    w = (torch.rand(n, n, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    ref = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    torch.matmul(a, w.t(), out=ref)
    runs = {"hipblaslt": lambda: torch.matmul(a, w.t(), out=ref)}
    outs = {}
    for v in VARIANTS:
        outs[v] = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        op = O.linear(a, w, outs[v], M=n, K=n, N=n, variant=v)
        runs[f"v{v}"] = (lambda op=op: O.launch(op))
    times = {k: [] for k in runs}
    for _ in range(ROUNDS):
        for k, fn in runs.items():
            times[k].append(timeit(fn))
    line = f"gemm {n}^3:"
    for k in runs:
        ts = sorted(times[k])
        line += f"  {k} {2.0 * n ** 3 / ts[len(ts) // 2] / 1e9:7.0f}"
    for v in VARIANTS:
        line += f"  |v{v}-ref| {float((outs[v].float() - ref.float()).abs().max()):.3g}"
    print(line, flush=True)
