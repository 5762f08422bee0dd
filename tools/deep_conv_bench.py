import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from marigold_amd import _lib as L, ops as O, weights as Wm
dev = torch.device("cuda:0"); L.init(0)
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
g = torch.Generator().manual_seed(1)
for (B, HW, Cin, N) in ((10, 12, 1280, 1280), (10, 12, 2560, 1280), (10, 24, 1280, 1280), (10, 24, 2560, 1280)):
    M = B * HW * HW
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = Wm.pack_conv3x3(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for v in [int(a) for a in sys.argv[1:]] or [0]:
        try:
            op = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=9, pad=1, bias=b, residual=r, variant=v)
            us = t(lambda: O.launch(op)); res.append(f"v{v}: {us:7.1f} us {2*M*N*9*Cin/us/1e6:6.0f} TF/s")
        except Exception as e:
            res.append(f"v{v}: ERR")
    print(f"conv3x3 {Cin}->{N} @{HW}x{HW} B={B}: " + "  ".join(res), flush=True)
