"""The collapsed cross-attention on the deep UNet levels: MG_OP_ROWGEMM's K-split form against the tile GEMM's MG_EPI_XATTN2."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marigold_amd import _lib as L, ops as O, weights as Wm
dev = torch.device("cuda:0"); L.init(0)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
g = torch.Generator().manual_seed(5)
for (B, T, C, heads) in ((10, 2304, 640, 10), (10, 576, 1280, 20), (10, 144, 1280, 20), (1, 576, 1280, 20)):
    M = B * T
    x = (torch.randn(M, C, generator=g) * 0.7).to(torch.bfloat16)
    st = torch.stack([x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous().to(dev)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ctx = torch.randn(2, 1024, generator=g)
    wq, wo = torch.randn(C, C, generator=g) / math.sqrt(C), torch.randn(C, C, generator=g) / math.sqrt(C)
    wk, wv = torch.randn(C, 1024, generator=g) / 32, torch.randn(C, 1024, generator=g) / 32
    bo = (0.1 * torch.randn(C, generator=g)).to(dev)
    wqk, vot, npad = Wm.cross_attention_tables(wq, wk, wv, wo, ctx, heads)
    wp, lg, lc = Wm.fold_layernorm(wqk, None, gamma, beta)
    pk = Wm.pack_rowgemm_xattn_ksplit(wp.float(), lc, lg, vot, bo).to(dev)
    sc = 1 / math.sqrt(C // heads)
    h1 = x.to(dev).clone(); so1 = torch.zeros(M, 2, device=dev)
    op1 = O.rowgemm(h1, pk, h1, M=M, K=C, N=64, form=L.RG_XATTN, ln_in=st, ln_out=so1, sm_cols=2 * heads, sm_scale=sc)
    O.launch(op1); torch.cuda.synchronize()
    r1, s1 = h1.clone(), so1.clone()
    h2 = x.to(dev).clone(); so2 = torch.zeros(M, 2, device=dev)
    wpd, lgd, lcd, votd = wp.to(dev), lg.to(dev), lc.to(dev), vot.to(dev, torch.bfloat16)
    op2 = O.linear(h2, wpd, h2, M=M, K=C, N=64, epi=L.EPI_XATTN2, ln_in=st, ln_g=lgd, ln_c=lcd, sm_scale=sc, sm_cols=2 * heads,
                   out2=votd, c2=C, ldo=C, bias=bo, residual=h2, ldr=C, ln_out=so2)
    O.launch(op2); torch.cuda.synchronize()
    d = (r1.float() - h2.float()).abs().max().item(); ds = (s1 - so2).abs().max().item(); nd = int((r1 != h2).sum())
    # fp64 reference on the first 512 rows: whose rounding is it?
    xs = x[:512].double()
    y = (xs - xs.mean(1, keepdim=True)) * (xs.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * gamma.double() + beta.double()
    q = (y @ wq.double().t()).view(-1, heads, C // heads)
    kk = (ctx.double() @ wk.double().t()).view(2, heads, C // heads); vv = (ctx.double() @ wv.double().t()).view(2, heads, C // heads)
    p = torch.softmax(torch.einsum("mhd,jhd->mhj", q, kk) * sc, dim=-1)
    ref = torch.einsum("mhj,jhd->mhd", p, vv).reshape(-1, C) @ wo.double().t() + bo.cpu().double() + xs
    e1 = ((r1[:512].double().cpu() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item()
    e2 = ((h2[:512].double().cpu() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item()
    b1 = (r1[:512].double().cpu() - ref).mean().item(); b2 = (h2[:512].double().cpu() - ref).mean().item()
    print(f"   vs fp64 reference (512 rows): K-split rms-rel {e1:.3e} mean err {b1:+.2e};  tile GEMM rms-rel {e2:.3e} mean err {b2:+.2e}")
    u1, u2 = t(lambda: O.launch(op1)), t(lambda: O.launch(op2))
    print(f"xattn B={B} T={T} C={C}: K-split {u1:6.1f} us   tile GEMM {u2:6.1f} us   max |d out| {d:.3e} (differing {nd} of {r1.numel()}), max |d stats| {ds:.3e}", flush=True)
