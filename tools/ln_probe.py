"""Cost of the folded LayerNorm on the consumer side: GEGLU / qkv-shaped Linear layers with and without ln_in."""
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
for (M, K, N, geglu) in ((92160, 320, 2560, 1), (23040, 640, 5120, 1), (5760, 1280, 10240, 1), (92160, 320, 640, 0), (23040, 640, 1280, 0)):
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
    st = torch.rand(M, 2, generator=g).to(dev)
    lg, lc = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    r = []
    for v in (0, 51, 33, 62):
        if v == 62 and N % 256: continue
        kw = dict(M=M, K=K, N=N, bias=b, epi=L.EPI_GEGLU if geglu else L.EPI_BF16, variant=v)
        o0 = O.linear(x, w, out, **kw)
        o1 = O.linear(x, w, out, ln_in=st, ln_g=lg, ln_c=lc, **kw)
        a, c = t(lambda: O.launch(o0)), t(lambda: O.launch(o1))
        r.append(f"v{v}: {a*1e3:.0f} / {c*1e3:.0f} us")
    print(f"M={M} K={K} N={N} {'geglu' if geglu else 'plain'} (no fold / folded LN): " + "   ".join(r), flush=True)
