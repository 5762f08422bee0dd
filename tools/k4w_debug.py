#!/usr/bin/env python
"""Debug aid for tile variant 72 (hand-placed K loop): where do wrong outputs sit - which 32x32 blocks, which k-steps?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
V = int(os.environ.get("V", "72"))


def run(x, w, M, K, N):
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    O.launch(O.linear(x, w, out, M=M, K=K, N=N, variant=V))
    torch.cuda.synchronize()
    return out.float().cpu()


g = torch.Generator().manual_seed(3)
for M, N, K in ((256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 256), (256, 256, 512), (512, 768, 1024)):
    x = (torch.rand(M, K, generator=g) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, generator=g) * 2 - 1).bfloat16()
    ref = x.float() @ w.float().t()
    xd, wd = x.to(dev), w.to(dev)
    outs = [run(xd, wd, M, K, N) for _ in range(3)]
    err = (outs[0] - ref).abs()
    print(f"M{M} N{N} K{K}: max err {float(err.max()):.3g} (scale {float(ref.abs().max()):.3g}); runs equal: "
          f"{torch.equal(outs[0], outs[1])} {torch.equal(outs[0], outs[2])}; nan {int(torch.isnan(outs[0]).sum())}")
    bm = err[:256, :256].reshape(8, 32, 8, 32).amax(dim=(1, 3))
    print("  32x32 block max err (rows = pixel blocks, cols = channel blocks):")
    for r in range(8):
        print("   " + " ".join(f"{float(v):6.2f}" for v in bm[r]))
    if K <= 512:
        bad = []
        for j in range(K // 16):
            xj = torch.zeros_like(x)
            xj[:, 16 * j:16 * j + 16] = x[:, 16 * j:16 * j + 16]
            o = run(xj.to(dev), wd, M, K, N)
            e = float((o - xj.float() @ w.float().t()).abs().max())
            if e > 0.05:
                bad.append((j, round(e, 2)))
        print(f"  wrong k-steps (16 wide): {bad}")
        # which source k does a wrong k-step show?  one-hot x column -> output = w[:, k']
        if bad:
            j = bad[0][0]
            xo = torch.zeros(M, K).bfloat16()
            xo[:, 16 * j] = 1.0
            o = run(xo.to(dev), wd, M, K, N)
            # match output row 0 against the columns of w
            d = (w.float().t().unsqueeze(0) - o[0].unsqueeze(0).unsqueeze(0)).abs().amax(dim=-1)[0]
            print(f"  one-hot k={16 * j}: row 0 matches w[:, k'] for k' = {[int(i) for i in torch.nonzero(d < 1e-3).flatten()][:8]}, "
                  f"rows equal to row 0: {int((o == o[0]).all(dim=1).sum())} of {M}")
