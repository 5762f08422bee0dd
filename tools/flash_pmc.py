#!/usr/bin/env python
"""A few launches of selected flash-attention variants at the level-0 shape, for rocprofv3 --pmc passes (scripts/gpu_pmc_flash.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
B, heads, T = 10, 5, 9216
C = heads * 64
g = torch.Generator(device="cpu").manual_seed(2)
qkv = torch.randn(B, T, 3 * C, generator=g).to(dev, torch.bfloat16)
vt = qkv[:, :, 2 * C:].permute(0, 2, 1).contiguous()
vtp = O.permute_vt_keys(vt)
out = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
ws = torch.zeros(O.FLASH_WS_BYTES, dtype=torch.uint8, device=dev)
for v in [int(x) for x in os.environ.get("FLASH_VARIANTS", "25,26").split(",")]:
    perm = v in (19, 20, 25, 26)
    op = O.flash_attn64(qkv, qkv[:, :, C:], vtp if perm else vt, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T,
                        sq=T * 3 * C, sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125, variant=v, vt_perm=perm,
                        ws=ws if v == 26 else None, ws_bytes=O.FLASH_WS_BYTES if v == 26 else 0)
    for _ in range(3):
        O.launch(op)
    torch.cuda.synchronize()
