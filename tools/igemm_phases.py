#!/usr/bin/env python
"""Where does a short-K MG_OP_IGEMM launch spend its time?  (tuning tool; run on the MI355X with MARIGOLD_TUNING=1
MARIGOLD_IGEMM_STAMPS=1)  Every workgroup of a forced-tile launch stamps the 100 MHz s_memrealtime at: kernel entry, operands
addressed, first K tile landed, K loop done, outputs stored, exit (csrc/igemm2_body.h::stamp; the instrumented instantiations of the
hand-placed tiles, variants 72 / 73 - the compiled tiles were stamped the same way for profiles/r5_igemm_phase_stamps.log).  Prints, per case, the launch's
wall time by HIP events and the distribution over workgroups of each phase and of the start / end times relative to the first
workgroup's entry."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

assert os.environ.get("MARIGOLD_TUNING") == "1" and os.environ.get("MARIGOLD_IGEMM_STAMPS") == "1"
dev = torch.device("cuda:0")
L.init(0)


def case(name, M, N, K, variant, taps=1, res=True, ln_out=False, HW=None, B=1, splits=1):
    g = torch.Generator(device="cpu").manual_seed(1)
    Cin = K // taps
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(dev, torch.bfloat16) if res else None
    st = torch.zeros(M * (N // 32 + 1) * 2, device=dev) if ln_out else None
    ctr = torch.zeros(65536, dtype=torch.int32, device=dev) if ln_out else None
    if taps == 1:
        op = O.linear(x, w, out, M=M, K=K, N=N, bias=bias, residual=r, ln_out=st, ln_counters=ctr, variant=variant, splits=splits)
    else:
        op = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=taps, stride=1, pad=1, bias=bias, residual=r,
                     variant=variant, splits=splits)
    for _ in range(3):
        O.launch(op)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        O.launch(op)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    O.launch(op)
    torch.cuda.synchronize()
    nwg_max = 8192
    buf = np.zeros(nwg_max * 8, dtype=np.uint64)
    L.check(L.load().mg_debug_read_workspace(buf.ctypes.data, buf.nbytes), "mg_debug_read_workspace")
    s = buf.reshape(nwg_max, 8).astype(np.int64)
    # workgroups of this launch: entries whose stamps are ordered and recent (the workspace is reused across cases)
    t_last = s[:, 5].max()
    live = (s[:, 0] > 0) & (s[:, 5] >= s[:, 0]) & (s[:, 5] > t_last - 100000)
    s = s[live]
    t0 = s[:, 0].min()
    ph = {"addr": s[:, 1] - s[:, 0], "first tile": s[:, 2] - s[:, 1], "K loop": s[:, 3] - s[:, 2], "epilogue": s[:, 4] - s[:, 3],
          "finish": s[:, 5] - s[:, 4], "start": s[:, 0] - t0, "end": s[:, 5] - t0, "whole": s[:, 5] - s[:, 0]}
    flops = 2.0 * M * N * K
    print(f"{name}: M={M} N={N} K={K} v{variant} s{splits}: {us:.1f} us/launch ({flops / us / 1e6:.0f} TFLOP/s), {len(s)} workgroups stamped")
    for k, v in ph.items():
        v = v / 100.0   # 100 MHz -> us
        print(f"    {k:10s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")


if __name__ == "__main__":
    case("gemm 4096^3 256x256", 4096, 4096, 4096, 72, res=False)
    case("lvl1 conv 1280->640 3x3", 23040, 640, 11520, 73, taps=9, HW=48, B=10, res=False)
    case("lvl2 conv 1280->1280 3x3 256x256", 5760, 1280, 11520, 72, taps=9, HW=24, B=10, res=False)
    case("lvl1 to_out (+res, +row stats)", 23040, 640, 640, 73, ln_out=True)
    case("lvl1 proj_out (+res)", 23040, 640, 640, 73)
    case("lvl0 ff.out (+res)", 92160, 320, 1280, 73)
    case("lvl1 ff.out (+res)", 23040, 640, 2560, 73)
    case("lvl2 proj_out (+res) 192x320", 5760, 1280, 1280, 73)
    case("lvl2 ff.out (+res) 256x256", 5760, 1280, 5120, 72)
    case("lvl1 conv2 3x3 (+res)", 23040, 640, 5760, 73, taps=9, HW=48, B=10)
