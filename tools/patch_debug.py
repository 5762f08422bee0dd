import math, os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O, weights as Wm
dev = torch.device("cuda:0"); L.init(0)
def run(B,H,W,C0,N,variant,reps=6):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C0, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(N, C0, 3, 3, generator=g) / math.sqrt(9*C0)).to(torch.bfloat16).float()
    ref = F.conv2d(x, w, None, padding=1).permute(0,2,3,1)
    a0 = x.permute(0,2,3,1).contiguous().to(dev, torch.bfloat16)
    wd = Wm.pack_conv3x3(w).to(dev, torch.bfloat16)
    outs=[]
    for r in range(reps):
        out = torch.full((B,H,W,N), float('nan'), device=dev, dtype=torch.bfloat16)
        O.launch(O.conv3x3(a0, wd, out, B=B,H=H,W=W,C0=C0,N=N,variant=variant)); torch.cuda.synchronize()
        outs.append(out.float().cpu())
    err=[(o-ref).abs().max().item() for o in outs]
    print(f"B{B} {H}x{W} C{C0} N{N} v{variant}: errs {['%.3e'%e for e in err]}")
    for r in range(1,reps):
        d=(outs[r]!=outs[0])
        if d.any():
            idx=d.nonzero()
            print(f"   run {r} differs at {d.sum().item()} elems; b {sorted(set(idx[:,0].tolist()))} y {sorted(set(idx[:,1].tolist()))[:20]} x {sorted(set(idx[:,2].tolist()))[:20]} n-range {idx[:,3].min().item()}..{idx[:,3].max().item()} nmod32 {sorted(set((idx[:,3]%32).tolist()))[:40]}; maxdiff {(outs[r]-outs[0]).abs().max().item():.3e}")
for cfg in [(2,16,16,64,256,1),(2,16,16,64,128,2),(2,16,16,64,128,4),(2,16,16,64,128,5),(2,16,16,64,320,3),(1,16,16,128,256,1),(1,32,32,64,256,1)]:
    run(*cfg)
