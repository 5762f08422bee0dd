"""MG_OP_ROWGEMM vs MG_OP_IGEMM on the token-local layers of the 96 x 96-token level (K = 320): correctness against an
fp32 torch reference of the same bf16 operands, then timing of both kernels, per layer form.

    python tools/rowgemm_bench.py [B] [waves...]
"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marigold_amd import _lib as L, ops as O, weights as Wm

dev = torch.device("cuda:0")
L.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
WAVES = [int(v) for v in sys.argv[2:]] or [12, 8]
NS = int(os.environ.get("ROWGEMM_NSPLIT", "0"))   # column split of the forms without row statistics
C = int(os.environ.get("ROWGEMM_C", "320"))
T = 9216 if C == 320 else 2304
M = B * T
g = torch.Generator().manual_seed(3)


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def rnd(*shape, s=1.0):
    return torch.randn(*shape, generator=g) * s


def err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9)), float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())


x = rnd(M, C, s=0.7).to(torch.bfloat16)
xd = x.to(dev)
mean = x.float().mean(1)
rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
st = torch.stack([mean, rstd], 1).contiguous().to(dev)
gamma, beta = 1 + 0.2 * rnd(C), 0.1 * rnd(C)
xs = x[: 4 * 384].float()                                     # reference rows (the first workgroups) + the last rows
xl = x[-768:].float()
lnf = lambda v: (v - v.mean(1, keepdim=True)) * (v.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * gamma + beta


def report(name, us_new, us_old, flops, byts, e):
    s = "  ".join(f"rowgemm/{w}w {u:7.1f} us ({flops / u / 1e6:6.0f} TF/s, {byts / u / 1e6:5.2f} TB/s)" for w, u in us_new.items())
    print(f"{name:12s} {s}   igemm {us_old:7.1f} us   max-rel {e[0]:.2e} rms-rel {e[1]:.2e}", flush=True)


# ---- QKV with the folded LayerNorm, V^T in the permuted key order -------------------------------------------------------
wq = rnd(3 * C, C, s=1 / math.sqrt(C))
wp, lg, lc = Wm.fold_layernorm(wq, None, gamma, beta)
pk = Wm.pack_rowgemm(wp.float(), lc, lg).to(dev)
ldt = T
qk = torch.zeros(M, 2 * C, device=dev, dtype=torch.bfloat16)
vt = torch.zeros(B, C, ldt, device=dev, dtype=torch.bfloat16)
us = {}
for w in WAVES:
    op = O.rowgemm(xd, pk, qk, M=M, K=C, N=3 * C, form=L.RG_QKV, ldo=2 * C, ln_in=st, vt=vt, tokens=T, ldt=ldt, trans_from=2 * C, waves=w, nsplit=NS)
    us[w] = t(lambda: O.launch(op))
qk2 = torch.zeros_like(qk); vt2 = torch.zeros_like(vt)
wp_d, lg_d, lc_d = wp.to(dev), lg.to(dev), lc.to(dev)      # (ops keep raw pointers: the tensors must outlive them)
op_old = O.igemm(xd, wp_d, qk2, B=B, H=T, W=1, Cin=C, Ho=T, Wo=1, N=3 * C, ldo=2 * C, out2=vt2, trans_from=2 * C, ldt=ldt,
                 ln_in=st, ln_g=lg_d, ln_c=lc_d, trans_perm=True)
us_old = t(lambda: O.launch(op_old))
ref = lambda v: lnf(v) @ wq.t()
r0, r1 = ref(xs), ref(xl)
e_qk = err(torch.cat([qk[: xs.shape[0]], qk[-768:]]), torch.cat([r0[:, : 2 * C], r1[:, : 2 * C]]))
vt_nat = vt[0].float().cpu().view(C, T // 16, 16)[:, :, torch.tensor(O.VT_PERM16).argsort()].reshape(C, T) if hasattr(O, "VT_PERM16") else None
e_v = err(vt_nat[:, : xs.shape[0]].t(), r0[:, 2 * C:]) if vt_nat is not None else (float("nan"),) * 2
same = bool((qk == qk2).all()) and bool((vt == vt2).all())
report("qkv+LN", us, us_old, 2 * M * 3 * C * C, (M * C + M * 3 * C) * 2, e_qk)
torch.cuda.synchronize()
print(f"             V^T max-rel {e_v[0]:.2e} rms-rel {e_v[1]:.2e}; bit-identical to igemm: {same}"
      f" (differing QK elements {int((qk != qk2).sum())}, V {int((vt != vt2).sum())}; max |d| {float((qk.float() - qk2.float()).abs().max()):.3e})", flush=True)

# ---- to_out: bias + residual in place + row statistics -----------------------------------------------------------------
wo, bo = rnd(C, C, s=1 / math.sqrt(C)), 0.1 * rnd(C)
pk = Wm.pack_rowgemm(wo, bo).to(dev)
h0 = rnd(M, C, s=1.0).to(torch.bfloat16)
us = {}
for w in WAVES:
    h = h0.to(dev).clone(); so = torch.zeros(M, 2, device=dev)
    op = O.rowgemm(xd, pk, h, M=M, K=C, N=C, residual=h, ln_out=so, waves=w)
    us[w] = t(lambda: O.launch(op))
h = h0.to(dev).clone(); so = torch.zeros(M, 2, device=dev)
O.launch(O.rowgemm(xd, pk, h, M=M, K=C, N=C, residual=h, ln_out=so, waves=WAVES[0])); torch.cuda.synchronize()
h2 = h0.to(dev).clone(); tab = torch.zeros(M * (C // 32) * 2 + M * 2, device=dev); so2 = tab[M * (C // 32) * 2:].view(M, 2); ctr = torch.zeros(65536, device=dev, dtype=torch.int32)
wo_d, bo_d = wo.to(dev, torch.bfloat16), bo.to(dev)
op_old = O.linear(xd, wo_d, h2, M=M, K=C, N=C, bias=bo_d, residual=h2, ln_out=tab, ln_counters=ctr)
O.launch(op_old); torch.cuda.synchronize()
so2 = so2.clone()
h3 = h0.to(dev).clone()
op_t = O.linear(xd, wo_d, h3, M=M, K=C, N=C, bias=bo_d, residual=h3, ln_out=tab, ln_counters=ctr)
us_old = t(lambda: O.launch(op_t))
r = xs @ wo.to(torch.bfloat16).float().t() + bo + h0[: xs.shape[0]].float()
e = err(h[: xs.shape[0]], r)
es = err(so[: xs.shape[0]], torch.stack([r.mean(1), (r.var(1, unbiased=False) + 1e-5).rsqrt()], 1))
report("to_out+res", us, us_old, 2 * M * C * C, 3 * M * C * 2, e)
print(f"             row statistics max-rel {es[0]:.2e}; vs igemm: out differing {int((h != h2).sum())}, stats max |d| {float((so - so2).abs().max()):.3e}")

# ---- proj_in with the GroupNorm apply folded into the load --------------------------------------------------------------
wi, bi = rnd(C, C, s=1 / math.sqrt(C)), 0.1 * rnd(C)
pk = Wm.pack_rowgemm(wi, bi).to(dev)
ss = torch.stack([1 + 0.3 * rnd(B, C), 0.2 * rnd(B, C)], 1).contiguous()      # [B][2][C]
ss_d, wi_d, bi_d = ss.to(dev), wi.to(dev, torch.bfloat16), bi.to(dev)
out = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
us = {}
for w in WAVES:
    op = O.rowgemm(xd, pk, out, M=M, K=C, N=C, gn_ss=ss_d, tokens=T, waves=w)
    us[w] = t(lambda: O.launch(op))
xn = torch.empty_like(xd); out2 = torch.zeros_like(out)
op_a = O.gn_apply(xd, ss_d, xn, B=B, HW=T, C=C, silu=False)
op_b = O.linear(xn, wi_d, out2, M=M, K=C, N=C, bias=bi_d)
us_old = t(lambda: (O.launch(op_a), O.launch(op_b)))
xr = torch.cat([(xs * ss[0, 0] + ss[0, 1]), (xl * ss[B - 1, 0] + ss[B - 1, 1])]).to(torch.bfloat16).float()
e = err(torch.cat([out[: xs.shape[0]], out[-768:]]), xr @ wi.to(torch.bfloat16).float().t() + bi)
report("gn+proj_in", us, us_old, 2 * M * C * C, 2 * M * C * 2, e)
print(f"             vs gn_apply + igemm: differing {int((out != out2).sum())}, max |d| {float((out.float() - out2.float()).abs().max()):.3e}")

# ---- GEGLU with the folded LayerNorm -------------------------------------------------------------------------------------
H = 4 * C
w1, b1 = rnd(2 * H, C, s=1 / math.sqrt(C)), 0.1 * rnd(2 * H)
order = Wm.rowgemm_geglu_order(2 * H)
wpo, lgo, lco = Wm.fold_layernorm(w1[order], b1[order], gamma, beta)
pk = Wm.pack_rowgemm(wpo.float(), lco, lgo).to(dev)
hid = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
if os.environ.get("ROWGEMM_DBG"):   # where a wave's time goes, per stage (s_memtime stamps: wait + barrier | MFMA phase | epilogue)
    for form, nm in ((L.RG_GEGLU, "geglu"),):
        dbg = torch.zeros(M // 32, 4, device=dev, dtype=torch.int64)
        O.launch(O.rowgemm(xd, pk, hid, M=M, K=C, N=2 * H, form=form, ln_in=st, dbg=dbg)); torch.cuda.synchronize()
        d = dbg.cpu().double()
        n = d[:, 3].clamp_min(1)
        print(f"   {nm}: per stage and wave, cycles: wait+barrier {float((d[:, 0] / n).mean()):7.0f}  issue+MFMA {float((d[:, 1] / n).mean()):7.0f}  epilogue {float((d[:, 2] / n).mean()):7.0f}"
              f"   (by wave of the workgroup, wait: {[int(v) for v in (d[:, 0] / n).view(-1, 12).mean(0)]}, MFMA: {[int(v) for v in (d[:, 1] / n).view(-1, 12).mean(0)]}, epi: {[int(v) for v in (d[:, 2] / n).view(-1, 12).mean(0)]})", flush=True)
us = {}
for w in WAVES:
    op = O.rowgemm(xd, pk, hid, M=M, K=C, N=2 * H, form=L.RG_GEGLU, ln_in=st, waves=w, nsplit=NS)
    us[w] = t(lambda: O.launch(op))
wg, bg = Wm.pack_geglu(w1, b1)
wpg, lgg, lcg = Wm.fold_layernorm(wg, bg, gamma, beta)
hid2 = torch.zeros_like(hid)
wpg_d, lgg_d, lcg_d = wpg.to(dev), lgg.to(dev), lcg.to(dev)
op_old = O.linear(xd, wpg_d, hid2, M=M, K=C, N=2 * H, epi=L.EPI_GEGLU, ln_in=st, ln_g=lgg_d, ln_c=lcg_d)
us_old = t(lambda: O.launch(op_old))
y = lnf(xs) @ w1.t() + b1
r = y[:, :H] * torch.nn.functional.gelu(y[:, H:])
e = err(hid[: xs.shape[0]], r)
report("geglu+LN", us, us_old, 2 * M * 2 * H * C, (M * C + M * H) * 2, e)
print(f"             vs igemm: differing {int((hid != hid2).sum())} of {hid.numel()}, max |d| {float((hid.float() - hid2.float()).abs().max()):.3e}")

# ---- collapsed cross-attention, in place --------------------------------------------------------------------------------
if C != 320:
    sys.exit(0)
heads = 5
ctx = rnd(2, 1024)
wq2, wo2 = rnd(C, C, s=1 / math.sqrt(C)), rnd(C, C, s=1 / math.sqrt(C))
wk2, wv2 = rnd(C, 1024, s=1 / 32), rnd(C, 1024, s=1 / 32)
wqk, vot, npad = Wm.cross_attention_tables(wq2, wk2, wv2, wo2, ctx, heads)
wpx, lgx, lcx = Wm.fold_layernorm(wqk, None, gamma, beta)
pk = Wm.pack_rowgemm_xattn(wpx.float(), lcx, lgx, vot, bo).to(dev)
us = {}
for w in [v for v in WAVES if v in (12, 8)]:
    hx = xd.clone(); so = torch.zeros(M, 2, device=dev)
    op = O.rowgemm(hx, pk, hx, M=M, K=C, N=64, form=L.RG_XATTN, ln_in=st, ln_out=so, sm_cols=2 * heads, sm_scale=1 / math.sqrt(C // heads), waves=w)
    us[w] = t(lambda: O.launch(op))
hx = xd.clone(); so = torch.zeros(M, 2, device=dev)
O.launch(O.rowgemm(hx, pk, hx, M=M, K=C, N=64, form=L.RG_XATTN, ln_in=st, ln_out=so, sm_cols=2 * heads, sm_scale=1 / math.sqrt(C // heads), waves=12))
torch.cuda.synchronize()
h2 = xd.clone(); so2 = torch.zeros(M, 2, device=dev)
wpx_d, lgx_d, lcx_d, vot_d = wpx.to(dev), lgx.to(dev), lcx.to(dev), vot.to(dev, torch.bfloat16)
op_old = O.linear(h2, wpx_d, h2, M=M, K=C, N=64, epi=L.EPI_XATTN2, ln_in=st, ln_g=lgx_d, ln_c=lcx_d, sm_scale=1 / math.sqrt(C // heads),
                  sm_cols=2 * heads, out2=vot_d, c2=C, ldo=C, bias=bo_d, residual=h2, ldr=C, ln_out=so2)
O.launch(op_old); torch.cuda.synchronize()
d = int((hx != h2).sum())
h3 = xd.clone()
op_t = O.linear(h3, wpx_d, h3, M=M, K=C, N=64, epi=L.EPI_XATTN2, ln_in=st, ln_g=lgx_d, ln_c=lcx_d, sm_scale=1 / math.sqrt(C // heads),
                sm_cols=2 * heads, out2=vot_d, c2=C, ldo=C, bias=bo_d, residual=h3, ldr=C, ln_out=so2)
us_old = t(lambda: O.launch(op_t))
report("xattn", us, us_old, 2 * M * C * 64 * 2, 2 * M * C * 2, (float("nan"), float("nan")))
print(f"             vs igemm XATTN2: differing {d} of {hx.numel()}, max |d| {float((hx.float() - h2.float()).abs().max()):.3e}, stats max |d| {float((so - so2).abs().max()):.3e}")
