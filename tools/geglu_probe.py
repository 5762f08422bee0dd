#!/usr/bin/env python
"""Probe of the short-K projection GEMMs (tuning tool): time vs K, N and epilogue, and the ablation tiles (40-42: no
LDS-DMA in the steady state / no MFMAs / no fragment reads - wrong results, timing only) on the 256x128 tile."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marigold_amd import _lib as L, ops as O
dev = torch.device("cuda:0"); L.init(0)
def t(fn, it=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
g = torch.Generator().manual_seed(1)
M = 92160
for (K, N) in ((64, 2560), (128, 2560), (320, 2560), (640, 2560), (320, 1280)):
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for v in (33, 40, 41, 42, 20, 26, 51, 21, 23):
        op = O.linear(x, w, out, M=M, K=K, N=N, bias=b, variant=v)
        res[v] = t(lambda: O.launch(op))
    print(f"K={K} N={N}: " + "  ".join(f"v{k} {v*1e3:6.0f}us" for k, v in res.items()), flush=True)
