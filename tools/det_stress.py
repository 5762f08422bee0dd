"""Run-to-run determinism of in-place Linear + residual launches (with / without the row-statistics output)."""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from marigold_amd import _lib as L, ops as O
dev = torch.device("cuda:0"); L.init(0)
g = torch.Generator().manual_seed(3)
for (M, K, N) in ((5760, 1280, 1280), (92160, 64, 320), (23040, 640, 640), (92160, 320, 320)):
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    h0 = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
    for variant in (35, 0, 32, 36, 51):
        for mode in ("ln_out", "plain"):
            first, nd = None, 0
            for it in range(10):
                h = h0.clone()
                st = torch.zeros((M * (N // 32 + 1), 2), device=dev)
                o = torch.empty_like(h) if "not-in-place" in mode else h
                O.launch(O.linear(a, w, o, M=M, K=K, N=N, bias=b, residual=h, ln_out=st if "ln_out" in mode else None, variant=variant))
                torch.cuda.synchronize()
                if first is None: first = o.clone()
                elif not torch.equal(first, o):
                    nd += 1
                    if nd == 1:
                        d = (first != o)
                        rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                        print(f"     rows {rows[:12].tolist()} ... ({rows.numel()}), cols {cols[:6].tolist()} .. {cols[-3:].tolist()} ({cols.numel()}), max diff {float((first.float()-o.float()).abs().max()):.3g}", flush=True)
            print(f"M={M} K={K} N={N} v{variant} {mode}: {nd}/9 runs differ", flush=True)
