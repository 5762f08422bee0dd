"""One launch each of the short-K Linear shapes per tile variant (run under rocprofv3 --pmc to count executed instructions
per wave: profiles/r2_shortk_pmc.log).  Usage: python tools/shortk_pmc.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marigold_amd import _lib as L, ops as O
dev = torch.device("cuda:0"); L.init(0)
g = torch.Generator().manual_seed(1)
M, N = 92160, 2560
for K in (64, 320):
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for v in (33, 21, 51, 53):
        O.launch(O.linear(x, w, out, M=M, K=K, N=N, variant=v))
        torch.cuda.synchronize()
