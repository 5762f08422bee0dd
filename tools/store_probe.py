import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from marigold_amd import _lib as L, ops as O
dev=torch.device("cuda:0"); L.init(0)
def t(fn,it=6):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
g=torch.Generator().manual_seed(1)
for (M,K,N) in ((92160,64,2560),(737280,64,320),(368640,64,640),(184320,64,1280),(46080,64,5120)):
    x=(torch.randn(M,K,generator=g)*0.5).to(dev,torch.bfloat16)
    w=(torch.randn(N,K,generator=g)/math.sqrt(K)).to(dev,torch.bfloat16)
    out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    r={}
    for v in (33,21,23,35):
        if N%128 and v in (33,21): 
            pass
        op=O.linear(x,w,out,M=M,K=K,N=N,variant=v)
        r[v]=t(lambda:O.launch(op))
    mb=M*N*2/1e6
    print(f"M={M} N={N} out {mb:.0f}MB: "+"  ".join(f"v{k} {v*1e3:5.0f}us ({mb/v/1e3:.2f} TB/s)" for k,v in r.items()),flush=True)
# reference: device copy of the same bytes
src=torch.empty(92160*2560,device=dev,dtype=torch.bfloat16); dst=torch.empty_like(src)
ms=t(lambda: dst.copy_(src)); print(f"torch copy 472MB: {ms*1e3:.0f}us ({472/ms/1e3:.2f} TB/s each way)")
ms=t(lambda: dst.zero_()); print(f"torch memset 472MB: {ms*1e3:.0f}us ({472/ms/1e3:.2f} TB/s)")
