"""The transposed (V^T) section of the fused QKV projection alone, per tile variant, with the folded LayerNorm."""
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
for (B, T, C) in ((10, 9216, 320), (10, 2304, 640), (10, 576, 1280)):
    M = B * T
    x = (torch.randn(M, C, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, torch.bfloat16)
    st = torch.rand(M, 2, generator=g).to(dev)
    lg, lc = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    ldt = (T + 63) // 64 * 64
    vt = torch.zeros(B, C, ldt, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    r = []
    for v in (0, 46, 35, 25, 32, 21, 33, 23):
        try:
            op = O.igemm(x, w, None, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=C, ldo=C, out2=vt, trans_from=0, ldt=ldt,
                         ln_in=st, ln_g=lg, ln_c=lc, variant=v)
            r.append(f"v{v}: {t(lambda: O.launch(op))*1e3:.0f}")
        except Exception as e:
            r.append(f"v{v}: ERR")
    op = O.linear(x, w, out, M=M, K=C, N=C, ln_in=st, ln_g=lg, ln_c=lc)
    print(f"V^T section B={B} T={T} C={C} (us): " + "  ".join(r) + f"   | same columns row-major (auto): {t(lambda: O.launch(op))*1e3:.0f}", flush=True)
