#!/usr/bin/env python
"""Interleaved timing of the patch-resident conv3x3 (MG_OP_CONV3X3) against the implicit GEMM on the dominant UNet /
VAE shapes at the benchmark batch: plain convolution, and the fused chain (GroupNorm apply + SiLU inside the conv)
against gn_apply + implicit GEMM.  Tuning tool, not part of the product path."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
PV = tuple(int(v) for v in os.environ.get("PATCH_VARIANTS", "1,2,3,4").split(","))


def timeit(fns, rounds=3, iters=4):
    ts = {k: [] for k in fns}
    for r in range(rounds):
        for k, fn in fns.items():
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / iters)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


def case(name, B, HW, Cin, N, subpix=False):
    g = torch.Generator().manual_seed(1)
    M = B * HW * HW
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    T = 4 if subpix else 9
    w = (torch.randn((4 if subpix else 1) * N, T * Cin, generator=g) / math.sqrt(T * Cin)).to(dev, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    ss = torch.stack([1 + 0.1 * torch.randn(B, Cin, generator=g), 0.1 * torch.randn(B, Cin, generator=g)], 1).to(dev).contiguous()
    Mo = M * (4 if subpix else 1)
    out = torch.empty(Mo, N, device=dev, dtype=torch.bfloat16)
    h = torch.empty(M, Cin, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * Mo * N * T * Cin / (4 if subpix else 1) * (4 if subpix else 1) / (4 if subpix else 1)
    fl = 2.0 * M * N * T * Cin * (4 if subpix else 1)
    fns = {}
    if subpix:
        ig = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=4, stride=1, pad=1, bias=bias, batch_z=4,
                     zstrides=(0, N * 4 * Cin, 0, 0))
        fns["igemm"] = lambda: O.launch(ig)
        for v in PV:
            if (v in (3, 6, 7, 11) and N % 320) or (v in (1, 9, 10) and N % 256) or (v == 8 and N % 128):
                continue
            op = O.conv3x3(x, w, out, B=B, H=HW, W=HW, C0=Cin, N=N, subpix=True, bias=bias, wz=N * 4 * Cin, variant=v)
            fns[f"p{v}"] = (lambda op=op: O.launch(op))
    else:
        ig = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=9, stride=1, pad=1, bias=bias)
        ap = O.gn_apply(x, ss, h, B=B, HW=HW * HW, C=Cin, silu=True)
        ig2 = O.igemm(h, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=9, stride=1, pad=1, bias=bias)
        fns["igemm"] = lambda: O.launch(ig)
        fns["apply+igemm"] = lambda: (O.launch(ap), O.launch(ig2))
        for v in PV:
            if (v in (3, 6, 7, 11) and N % 320) or (v in (1, 9, 10) and N % 256) or (v == 8 and N % 128):
                continue
            op = O.conv3x3(x, w, out, B=B, H=HW, W=HW, C0=Cin, N=N, bias=bias, variant=v)
            opf = O.conv3x3(x, w, out, B=B, H=HW, W=HW, C0=Cin, N=N, bias=bias, ss=ss, silu=True, variant=v)
            fns[f"p{v}"] = (lambda op=op: O.launch(op))
            fns[f"p{v}+gn"] = (lambda op=opf: O.launch(op))
    ms = timeit(fns)
    print(f"{name:30s} M={M:8d} N={N:5d} Cin={Cin:5d} | " + "  ".join(f"{k}: {ms[k] * 1e3:7.1f}us {fl / ms[k] / 1e9:5.0f}TF" for k in ms), flush=True)


E = 10
CASES = os.environ.get("PATCH_CASES", "")
for c in [c for c in [("unet 320->320 @96", E, 96, 320, 320), ("unet 640->320 @96", E, 96, 640, 320), ("unet 960->320 @96", E, 96, 960, 320),
          ("unet 640->640 @48", E, 48, 640, 640), ("unet 1280->640 @48", E, 48, 1280, 640),
          ("vae 512->512 @96", E, 96, 512, 512), ("vae 512->512 @192 B4", 4, 192, 512, 512), ("vae 256->256 @384 B4", 4, 384, 256, 256),
          ("vae 128->128 @768 B2", 2, 768, 128, 128), ("vae 256->128 @768 B2", 2, 768, 256, 128)] if CASES in c[0]]:
    try:
        case(*c)
    except Exception as e:  # noqa: BLE001
        print(f"{c[0]}: FAILED {type(e).__name__}: {e}", flush=True)
for c in [c for c in [("up 640->640 @48->96", E, 48, 640, 640), ("up vae 512 @192->384 B4", 4, 192, 512, 512), ("up vae 256 @384->768 B2", 2, 384, 256, 256)] if CASES in c[0]]:
    try:
        case(*c, subpix=True)
    except Exception as e:  # noqa: BLE001
        print(f"{c[0]}: FAILED {type(e).__name__}: {e}", flush=True)
