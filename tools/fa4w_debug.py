#!/usr/bin/env python
"""Where does flash_attn64 variant 26 differ from SDPA?  Error map by query block / channel block.  Debug tool."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402
dev = torch.device("cuda:0")
L.init(0)
T = int(os.environ.get("T", "256"))
mode = os.environ.get("MODE", "rand")
C = 64
g = torch.Generator().manual_seed(1)
q = torch.randn(1, T, C, generator=g).bfloat16().float()
k = torch.randn(1, T, C, generator=g).bfloat16().float()
v = torch.randn(1, T, C, generator=g).bfloat16().float()
if mode == "vones":
    v = torch.ones_like(v)
if mode == "kzero":
    k = torch.zeros_like(k)       # uniform attention: out = mean of v
if mode == "vtile":               # v = index of its key tile: out = attention mass per tile
    v = (torch.arange(T) // 64).float()[None, :, None].expand(1, T, C).contiguous()
ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
qkv = torch.cat([q, k], dim=-1).to(dev, torch.bfloat16)
vtp = O.permute_vt_keys(v.permute(0, 2, 1).contiguous().to(dev, torch.bfloat16))
for var in (25, 26):
    out = torch.full((1, T, C), float("nan"), device=dev, dtype=torch.bfloat16)
    O.launch(O.flash_attn64(qkv, qkv[:, :, C:], vtp, out, B=1, heads=1, Ntok=T, ldq=2 * C, ldo=C, ldvt=T, sq=0, sk=0, svt=0, so=0,
                            scale=0.125, variant=var, vt_perm=True))
    torch.cuda.synchronize()
    o = out.float().cpu()[0]
    e = (o - ref[0]).abs()
    e[~torch.isfinite(e)] = 1e9
    print(f"variant {var} T={T} mode={mode}: max err {e.max():.3e}; bad (>0.05) {int((e > 0.05).sum())} of {e.numel()}")
    if var == 26:
        bad = e > 0.05
        print(" bad per 32-query block:", [int(bad[i:i + 32].sum()) for i in range(0, T, 32)][:16])
        print(" bad per channel:", [int(bad[:, c].sum()) for c in range(C)])
        print(" bad per query mod 32:", [int(bad[i::32].sum()) for i in range(32)])
        r = int(torch.nonzero(bad.any(1))[0]) if bad.any() else 0
        print(f" row {r}: got", [f"{x:.3g}" for x in o[r, :12].tolist()], "ref", [f"{x:.3g}" for x in ref[0, r, :12].tolist()])
