"""Where an ensembling cost evaluation's time goes on the GPU box: the device pass (launch + synchronise), the host arithmetic,
scipy's BFGS / line-search bookkeeping."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marigold_amd import _lib as L, ensemble as E
from scipy.optimize import minimize
L.init(0)
g = torch.Generator().manual_seed(3)
base = torch.rand(1, 1, 768, 768, generator=g)
d = (base * (1 + 0.1 * torch.rand(10, 1, 1, 1, generator=g)) + 0.05 * torch.rand(10, 1, 768, 768, generator=g)).cuda()
al = E.DepthAligner(d.float(), True, True, "median", 0.02)
p = al.init_param()
s32, t32 = p[:10].astype(np.float32), p[10:].astype(np.float32)
for _ in range(20): al.backend.regulariser(s32, t32)
N = 300
t0 = time.perf_counter()
for _ in range(N): al.backend.regulariser(s32, t32)
t_reg = (time.perf_counter() - t0) / N * 1e6
t0 = time.perf_counter()
for _ in range(N): al.cost_and_grad(p)
t_cg = (time.perf_counter() - t0) / N * 1e6
cnt = [0]
def f(q):
    cnt[0] += 1
    return al.reference_fd_objective(q)
t0 = time.perf_counter(); r = minimize(f, p, jac=True, method="BFGS", tol=1e-6, options={"maxiter": 50}); dt = time.perf_counter() - t0
print(f"device pass (launch + sync) {t_reg:.1f} us; cost_and_grad (host arithmetic + device pass) {t_cg:.1f} us; "
      f"scipy BFGS: {cnt[0]} evaluations, {r.nit} iterations, {dt * 1e3:.2f} ms = {dt / cnt[0] * 1e6:.1f} us per evaluation")
