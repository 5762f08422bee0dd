#!/usr/bin/env python
"""Runs a handful of representative launches (flash attention at 9216 tokens, the dominant UNet / VAE
convolutions, a GEGLU projection) a few times each, for rocprofv3 --pmc passes (MFMA utilisation,
wave stall reasons) - see scripts/gpu_pmc.sh.  Not part of the product path."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
g = torch.Generator(device="cpu").manual_seed(1)
REPS = int(os.environ.get("REPS", "3"))


def conv(B, HW, Cin, N, taps, geglu=False, variant=0):
    M = B * HW * HW
    K = taps * Cin
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
    op = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=taps, stride=1,
                 pad=1 if taps == 9 else 0, bias=bias, epi=L.EPI_GEGLU if geglu else L.EPI_BF16, variant=variant)
    for _ in range(REPS):
        O.launch(op)
    torch.cuda.synchronize()


def flash(B, heads, T):
    C = heads * 64
    qkv = torch.randn(B, T, 2 * C, generator=g).to(dev, torch.bfloat16)
    ldvt = (T + 63) // 64 * 64
    vt = torch.randn(B, C, ldvt, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    op = O.flash_attn64(qkv, qkv.data_ptr() + C * 2, vt, out, B=B, heads=heads, Ntok=T, ldq=2 * C, ldo=C,
                        ldvt=ldvt, sq=T * 2 * C, sk=T * 2 * C, svt=C * ldvt, so=T * C, scale=0.125)
    for _ in range(REPS):
        O.launch(op)
    torch.cuda.synchronize()


flash(10, 5, 9216)
conv(10, 96, 320, 320, 9)      # 128x64 tile
conv(10, 48, 640, 640, 9)      # 256x128 tile, 3 stages
conv(10, 96, 512, 512, 9)      # 256x256 tile, ping-pong K loop (VAE; the automatic choice)
conv(10, 96, 512, 512, 9, variant=34)   # same tile, one barrier per K tile
conv(10, 96, 320, 2560, 1, geglu=True)
print("microbench done")
