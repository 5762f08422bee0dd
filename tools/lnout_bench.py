#!/usr/bin/env python
"""Cost of the row-statistics hand-off (ln_out) in the tile GEMM's epilogue at the UNet's projection shapes: the same GEMM with
and without it, per variant.  Tuning tool, not product path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)


def timeit(fn, warm=2, iters=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(M, K, N, variants):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ln = torch.zeros(M * (N // 32) + M, 2, device=dev, dtype=torch.float32)
    line = f"M={M} K={K} N={N}:"
    for v in variants:
        t = {}
        for name, kw in (("plain", {}), ("ln_out", {"ln_out": ln})):
            op = O.igemm(a, w, out, B=1, H=M, W=1, Cin=K, Ho=M, Wo=1, N=N, residual=res, variant=v, **kw)
            try:
                t[name] = timeit(lambda: O.launch(op))
            except Exception as e:  # noqa: BLE001
                t[name] = float("nan")
        line += f"  v{v}: {t['plain']:.1f} / {t['ln_out']:.1f} us"
    print(line, flush=True)


if __name__ == "__main__":
    case(23040, 640, 640, (0, 73, 46, 51, 32))
    case(5760, 1280, 1280, (0, 62, 51, 32))
    case(23040, 2560, 640, (0, 73))
