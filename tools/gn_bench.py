#!/usr/bin/env python
"""GroupNorm launch forms timed in isolation on one MI355X: the statistics + apply pair against the one-launch slab form, at the
benchmark's shapes (round 6 also timed a one-launch ticket-barrier form here: profiles/r6_gn_coop_lost.log).  Each form is timed back to back with itself (HIP events, median of 5 x 20 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marigold_amd import _lib as L, ops as O

dev = torch.device("cuda", 0)
L.init(0)

def timeit(fns, iters=20, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            for f in fns:
                f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]

shapes = [(10, 2304, 640, 0), (10, 2304, 1280, 640), (10, 576, 1280, 0), (10, 576, 1280, 1280), (10, 9216, 320, 0), (10, 9216, 512, 0), (1, 9216, 512, 0), (10, 144, 1280, 0)]
for B, HW, C0, C1 in shapes:
    C = C0 + C1
    x0 = torch.randn(B, HW, C0, device=dev).bfloat16()
    x1 = torch.randn(B, HW, C1, device=dev).bfloat16() if C1 else None
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty(B, HW, C, device=dev, dtype=torch.bfloat16)
    ss = torch.empty(B, 2, C, device=dev)
    res = {}
    sch = max(1, min(HW // 32, 64, max(8, 288 // B)))
    nsrc = 2 if C1 else 1
    part2 = torch.empty(B, sch * nsrc, 32, 2, device=dev)
    ctr = torch.zeros(1024, dtype=torch.int32, device=dev)
    st = O.gn_stats(x0, part2, B=B, HW=HW, C=C0, chunks=sch, groups=32, Ctot=C, coff=0, slot0=0, slots=sch * nsrc, gamma=gamma, beta=beta, ss=ss, counters=ctr, eps=1e-5, x1=x1, C1=C1)
    ap = O.gn_apply(x0, ss, out, B=B, HW=HW, C=C, silu=True, x1=x1, C0=C0)
    res["stats"] = timeit([lambda: O.launch(st)])
    res["apply"] = timeit([lambda: O.launch(ap)])
    res["stats+apply"] = timeit([lambda: O.launch(st), lambda: O.launch(ap)])
    try:
        sl = O.gn_slab(x0, out, ss, B=B, HW=HW, C=C, groups=32, gamma=gamma, beta=beta, eps=1e-5, silu=True, x1=x1, C0=C0)
        res["slab"] = timeit([lambda: O.launch(sl)])
    except Exception as e:
        res["slab"] = None
    print(f"B={B} HW={HW} C={C0}+{C1} ({B*HW*C*2/1e6:.1f} MB): " + "  ".join(f"{k}={v:.1f}us" if v else f"{k}=n/a" for k, v in res.items()), flush=True)
