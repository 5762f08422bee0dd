import os, sys, torch
sys.path.insert(0, "/root/repo")
from marigold_amd import _lib as L, ops as O
dev = torch.device("cuda:0"); L.init(0)
def t(fn, it=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for (B, HW, C) in ((10, 9216, 320), (10, 9216, 640), (10, 2304, 640), (10, 2304, 1280), (10, 589824, 128), (1, 9216, 320)):
    x = torch.randn(B, HW, C).to(dev, torch.bfloat16)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = []
    for chunks in [c for c in (int(v) for v in os.environ.get('GN_CHUNKS', '12,16,24,32,40,51,76').split(',')) if c <= HW // 32]:
        part = torch.zeros(B * chunks * 32 * 2, device=dev); ss = torch.zeros(B * 2 * C, device=dev); ctr = torch.zeros(max(B, 1024), device=dev, dtype=torch.int32)
        op = O.gn_stats(x, part, B=B, HW=HW, C=C, chunks=chunks, groups=32, Ctot=C, coff=0, slot0=0, slots=chunks, gamma=gam, beta=bet, ss=ss, counters=ctr, eps=1e-5)
        res.append(f"chunks {chunks:4d}: {t(lambda: O.launch(op)):6.1f} us")
    print(f"gn_stats B={B} HW={HW} C={C} ({B*HW*C*2/1e6:.0f} MB): " + "  ".join(res), flush=True)
