#!/usr/bin/env python
"""One-process check of the experimental halo-shared 3x3 tile (variants 70-73) on the GPU: parity on a few shapes
against a validated tile, then interleaved timing against the automatic choice on the dominant UNet / VAE
convolutions.  Not part of the product path."""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O, weights as Wm  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
T0 = time.time()
BUDGET = float(os.environ.get("HALO_BUDGET_S", "20"))


def parity(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = (torch.randn(B, H, W, Cin, generator=g)).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(dev, torch.bfloat16)
    b = torch.randn(Cout, generator=g).to(dev)
    outs = []
    for v in (23, 70, 70, 71, 71, 72, 73):
        out = torch.full((B * H * W, Cout), float("nan"), device=dev, dtype=torch.bfloat16)
        O.launch(O.igemm(x, w, out, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=Cout, taps=9, stride=1, pad=1, bias=b, variant=v))
        torch.cuda.synchronize()
        outs.append(out.float())
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    e23, e70, e71 = ((o - ref).abs().max().item() for o in (outs[0], outs[1], outs[3]))
    e72, e73 = ((o - ref).abs().max().item() for o in (outs[5], outs[6]))
    same = torch.equal(outs[1], outs[2]) and torch.equal(outs[3], outs[4]) and torch.equal(outs[1], outs[3])
    print(f"parity B{B} {H}x{W} {Cin}->{Cout}: |err| v23 {e23:.3e} v70 {e70:.3e} v71 {e71:.3e} v72 {e72:.3e} v73 {e73:.3e} (scale {ref.abs().max().item():.2f}) "
          f"finite {bool(torch.isfinite(outs[1]).all() and torch.isfinite(outs[3]).all())} repeatable+identical {same}", flush=True)


def timing(name, B, HW, Cin, N, variants=(0, 33, 70, 71, 72, 73), rounds=3):
    M, K = B * HW * HW, 9 * Cin
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops = {v: O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=9, stride=1, pad=1, variant=v) for v in variants}
    ts = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            O.launch(ops[v])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                O.launch(ops[v])
            e1.record()
            torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * M * N * K
    print(f"{name:28s} M={M:7d} N={N:5d} K={K:6d} | " + " ".join(f"v{v}: {fl / sorted(ts[v])[len(ts[v]) // 2] / 1e9:6.0f}" for v in variants),
          flush=True)


for c in ((1, 12, 20, 64, 128), (3, 9, 31, 128, 192), (2, 40, 3, 64, 64), (2, 16, 24, 192, 320)):
    try:
        parity(*c)
    except Exception as e:  # noqa: BLE001
        print(f"parity {c}: FAILED {type(e).__name__}: {e}", flush=True)
for c in (("unet.conv 640->640 @48", 10, 48, 640, 640), ("unet.conv 1280->640 @48", 10, 48, 1280, 640),
          ("unet.conv 1280->1280 @24", 10, 24, 1280, 1280), ("unet.conv 640->320 @96", 10, 96, 640, 320),
          ("vae.conv 512->512 @96", 10, 96, 512, 512)):
    if time.time() - T0 > BUDGET:
        print("time budget reached", flush=True)
        break
    try:
        timing(*c)
    except Exception as e:  # noqa: BLE001
        print(f"timing {c[0]}: FAILED {type(e).__name__}: {e}", flush=True)
