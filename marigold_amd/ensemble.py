"""Test-time ensembling on the GPU - same functions, arguments, defaults and errors as the
reference's marigold/util/ensemble.py (``ensemble_depth`` :39-196, ``ensemble_normals``
:199-249), with every per-pixel pass a HIP kernel (csrc/ensemble.hip).

What the reference's optimiser actually does (and what is reproduced here)
--------------------------------------------------------------------------
``compute_param`` (:154-173) hands ``scipy.optimize.minimize(BFGS, tol=1e-6, maxiter=50)`` a cost
with NO gradient, so scipy takes forward differences with the absolute step h = 2^-26 = 1.49e-8
- but the cost first casts the parameters to the fp32 depth dtype (``torch.from_numpy(s).to(depth)``,
:109-114).  A perturbed parameter therefore only changes the cost if ``fp32(p_i + h) != fp32(p_i)``:
the finite difference equals  dcost/dp_i * k_i,  k_i = (fp32(p_i + h) - fp32(p_i)) / h.  For the
scales (s_i = 1/range_i >= 1, fp32 spacing 1.2e-7 >> h, init values on the fp32 grid) k_i == 0:
**the reference never moves the scales off ``init_param``; it optimises the shifts only**, and
those only while |t_i| is small enough for h to survive the rounding ([probed]: on every golden
case s_final / s_init == 1 exactly).  Taken literally the cost has a degenerate global optimum
(all s -> 0 costs just the 0.02 regulariser), so an "exact" optimiser collapses the ensemble; the
reference is only meaningful because of this quantisation.  This module keeps it: the BFGS sees
f(fp32(p)) and the analytic gradient multiplied by k (== the reference's finite differences in exact
arithmetic), without the reference's fp32 summation noise.  Result: identical scales, shifts
optimised at least as far as the reference gets (its noise stops it early), deterministic output.

How it runs on the device
-------------------------
* the E(E-1)/2 pairwise-RMSE reductions of the cost (:142-144, one ``.item()`` sync each in the
  reference) are a closed form of the per-member means and the centred E x E second-moment
  matrix, gathered ONCE per call by ``MG_OP_ENS_DEPTH_STATS``;
* the only per-evaluation pixel work is the fused align -> median -> min/max kernel of the
  regulariser (:146-150, ``MG_OP_ENS_DEPTH_MEDIAN``); it also reports the raw member values at the
  extremal pixels, from which the host forms the exact (sub)gradient - one kernel pair per BFGS
  evaluation instead of 2E+1 cost evaluations x (E(E-1)/2 + 2) host syncs;
* final align -> median (+MAD) -> normalise never leaves the device.
"""
import os

import numpy as np
import torch

from . import _lib as L, ops as O

_SCRATCH_BYTES = 12288
_FD_STEP = 1.4901161193847656e-08   # scipy's forward-difference step (sqrt(eps)), absolute


NATIVE_BFGS = os.environ.get("MARIGOLD_ENS_NATIVE_BFGS", "1") != "0"   # 0: scipy drives the alignment (A/B, parity of the native optimiser)
# Any ensemble size, like the reference (marigold/util/ensemble.py:39-49; script/depth/run.py:143-144 only warns above 15): the
# per-pixel order statistics hold <= 32 members in registers, 33 ... 128 in LDS, and beyond that select bitwise over the members
# in memory (csrc/ensemble.hip::depth_median_big_kernel); ``ensemble_normals`` loops over the members.
MAX_ENSEMBLE_SIZE = None


def _check_depth_args(depth, reduction, scale_invariant, shift_invariant):
    if depth.dim() != 4 or depth.shape[1] != 1:
        raise ValueError(f"Expecting 4D tensor of shape [B,1,H,W]; got {depth.shape}.")
    if reduction not in ("mean", "median"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    if not scale_invariant and shift_invariant:
        raise ValueError("Pure shift-invariant ensembling is not supported.")


def _q32(p):
    """The fp32 cast the reference applies to the parameters inside its cost (:109-114)."""
    return np.asarray(p, dtype=np.float64).astype(np.float32).astype(np.float64)


def fd_survival(p):
    """k_i = (fp32(p_i + h) - fp32(p_i)) / h: how much of scipy's forward-difference step survives
    the reference's fp32 parameter cast (0 for |p_i| >~ 0.25, i.e. always for the scales)."""
    p = np.asarray(p, dtype=np.float64)
    return (_q32(p + _FD_STEP) - _q32(p)) / _FD_STEP


class HipStatsBackend:
    """The two device passes the aligner needs, as HIP kernels."""

    def __init__(self, d, reduction, affine):
        E, HW = d.shape
        self.d, self.E, self.HW = d, E, HW
        self.red, self.affine = reduction, affine
        dev = d.device
        self.scratch = torch.empty(_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        # The optimiser calls the regulariser pass tens of times per map and each call moves 2E floats in and 2 + 2E out:
        # both live in pinned host memory that the kernels address directly (zero-copy over PCIe), so an evaluation is
        # one launch + one stream synchronisation instead of an H2D copy, a launch and a synchronising D2H copy.
        self.mm = torch.empty(2 + 2 * E, dtype=torch.float32, pin_memory=True)
        self.st = torch.empty(2 * E, dtype=torch.float32, pin_memory=True)
        self._mm_np, self._st_np = self.mm.numpy(), self.st.numpy()
        # every evaluation launches the same op on the same buffers: built once
        self._reg_op = O.ens_depth_median(self.d, self.st, None, None, self.mm, self.scratch, E=E, HW=HW,
                                          reduction=self.red, has_shift=self.affine)

    def stats(self):
        E = self.E
        dev = self.d.device
        # partial table: one (E x (E + 3)) block of doubles per pixel block the launcher uses (csrc/ensemble.hip: 128, 32 beyond 256
        # members, never more than the map has 256-pixel blocks) - at E = 1000 that is 257 MB where the fixed 128 asked for 1 GB
        nblk = min(-(-self.HW // 256), 32 if E > 256 else 128)
        sscratch = torch.empty(nblk * E * (E + 3), dtype=torch.float64, device=dev)
        stats = torch.empty(3 * E + E * E, dtype=torch.float64, device=dev)
        O.launch(O.ens_depth_stats(self.d, sscratch, stats, E=E, HW=self.HW))
        st = stats.cpu().numpy()
        return st[:E], st[E:2 * E], st[2 * E:3 * E], st[3 * E:].reshape(E, E)

    def regulariser(self, s32, t32):
        """(min, max of the ensembled aligned prediction, raw member values at those two pixels)."""
        E = self.E
        self._st_np[:E] = s32
        self._st_np[E:] = t32
        O.launch(self._reg_op)
        torch.cuda.current_stream(self.d.device).synchronize()
        r = self._mm_np.astype(np.float64)
        return r[0], r[1], r[2:2 + E], r[2 + E:]


class DepthAligner:
    """Cost / gradient of the reference's alignment objective for one stack of members
    (``backend`` supplies the two pixel passes; the default runs them as HIP kernels)."""

    def __init__(self, d32, scale_invariant, shift_invariant, reduction, regularizer_strength, backend=None):
        E = d32.shape[0]
        self.E, self.HW = E, d32.shape[2] * d32.shape[3]
        self.affine = scale_invariant and shift_invariant
        self.red = 0 if reduction == "median" else 1
        self.lam = float(regularizer_strength)
        self.backend = backend if backend is not None else HipStatsBackend(
            d32.reshape(E, self.HW).contiguous(), self.red, self.affine)
        self.dmin, self.dmax, self.mean, self.C = self.backend.stats()
        self.n_eval = 0

    def init_param(self):
        lo = self.dmin.astype(np.float32)
        hi = self.dmax.astype(np.float32)
        if self.affine:
            s = (np.float32(1.0) / np.maximum(hi - lo, np.float32(1e-6))).astype(np.float32)
            return np.concatenate([s, -s * lo]).astype(np.float64)
        return (np.float32(1.0) / np.maximum(hi, np.float32(1e-6))).astype(np.float64)

    def _split(self, p):
        if self.affine:
            return p[:self.E], p[self.E:]
        return p, np.zeros(self.E)

    def cost_and_grad(self, p):
        """Exact cost and gradient at the fp32-cast parameters.  The E x E arithmetic is one call into the library
        (``mg_ens_align_cost_grad``: the numpy form below, same operations in the same order - numpy's pairwise summation
        included - hence the same bits; as ~35 numpy calls on 10 x 10 operands it was 50-90 us of the ~130 us an
        evaluation costs, ~100 evaluations per map)."""
        self.n_eval += 1
        E = self.E
        k = self._consts()
        s, t = self._split(_q32(p))
        s = np.ascontiguousarray(s, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        out = k["out"]
        L.check(k["fn"](E, s.ctypes.data, t.ctypes.data, k["mean_p"], k["C_p"], out.ctypes.data, out.ctypes.data + 8,
                        out.ctypes.data + 8 * (1 + E)), "mg_ens_align_cost_grad")
        cost = float(out[0])
        gs, gt = out[1:1 + E].copy(), out[1 + E:1 + 2 * E].copy()
        return self._add_regulariser(cost, gs, gt, s, t)

    def cost_and_grad_numpy(self, p):
        """The same objective as numpy array operations (the form ``mg_ens_align_cost_grad`` restates; tests compare the
        two bit for bit)."""
        self.n_eval += 1
        k = self._consts()
        s, t = self._split(_q32(p))
        u = s * self.mean + t
        C, dC = self.C, k["dC"]
        s2d = (s * s) * dC
        du = u[:, None] - u[None, :]
        q = s2d[:, None] + s2d[None, :] - 2.0 * np.outer(s, s) * C + du * du
        np.maximum(q, 0.0, out=q)
        r = np.sqrt(q)
        cost = float(r[k["iu"]].sum())
        w = np.zeros_like(r)
        np.divide(0.5, r, out=w, where=r > 0)
        w[k["di"]] = 0.0
        # d q_ij / d s_i = 2 s_i C_ii - 2 s_j C_ij + 2 (u_i - u_j) m_i ; d q_ij / d t_i = 2 (u_i - u_j)
        gs = (w * (2.0 * (s * dC)[:, None] - 2.0 * s[None, :] * C + 2.0 * du * self.mean[:, None])).sum(axis=1)
        gt = (w * 2.0 * du).sum(axis=1)
        return self._add_regulariser(cost, gs, gt, s, t)

    def _add_regulariser(self, cost, gs, gt, s, t):
        E = self.E
        if self.lam > 0:
            s32, t32 = s.astype(np.float32), t.astype(np.float32)
            mn, mx, dmn, dmx = self.backend.regulariser(s32, t32)
            cost += (abs(0.0 - mn) + abs(1.0 - mx)) * self.lam
            for val, draw, sign in ((mn, dmn, np.sign(mn)), (mx, dmx, -np.sign(1.0 - mx))):
                a = (draw.astype(np.float32) * s32 + t32)
                if self.red == 0:  # lower-middle median: which member is it at that pixel?
                    order = np.argsort(a, kind="stable")
                    e = order[(E - 1) // 2]
                    gs[e] += self.lam * sign * draw[e]
                    gt[e] += self.lam * sign
                else:
                    gs += self.lam * sign * draw / E
                    gt += self.lam * sign / E
        g = np.concatenate([gs, gt]) if self.affine else gs
        return cost, g

    def _consts(self):
        k = getattr(self, "_k", None)
        if k is None:
            lib = L.load()
            mean = np.ascontiguousarray(self.mean, dtype=np.float64)
            C = np.ascontiguousarray(self.C, dtype=np.float64)
            k = self._k = {"dC": np.diag(self.C).copy(), "iu": np.triu_indices(self.E, 1), "di": np.diag_indices(self.E),
                           "fn": lib.mg_ens_align_cost_grad, "mean": mean, "C": C, "mean_p": mean.ctypes.data,
                           "C_p": C.ctypes.data, "out": np.zeros(1 + 2 * self.E)}
        return k

    def cost(self, p):
        return self.cost_and_grad(p)[0]

    def minimize_native(self, p0, tol, max_iter):
        """scipy.optimize.minimize(self.reference_fd_objective, p0, jac=True, method="BFGS", tol=tol, maxiter=max_iter) as ONE
        library call (mg_ens_align_minimize): -> (parameters, cost, iterations)."""
        import ctypes
        k = self._consts()
        b = self.backend
        x = np.array(p0, dtype=np.float64).copy()
        fval, nit, nfev, status = ctypes.c_double(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        stream = torch.cuda.current_stream(b.d.device).cuda_stream
        L.check(L.load().mg_ens_align_minimize(ctypes.addressof(b._reg_op), ctypes.c_void_p(stream), self.E, int(self.affine), int(self.red),
                                               float(self.lam), k["mean_p"], k["C_p"], b._st_np.ctypes.data, b._mm_np.ctypes.data,
                                               x.ctypes.data, float(tol), int(max_iter), ctypes.byref(fval), ctypes.byref(nit),
                                               ctypes.byref(nfev), ctypes.byref(status)), "mg_ens_align_minimize")
        self.n_eval += nfev.value
        self.status = status.value   # scipy's warnflag: 0 converged, 1 maxiter, 2 precision loss, 3 NaN
        if status.value == 3 or not np.all(np.isfinite(x)):
            import warnings
            warnings.warn("ensemble_depth: the alignment optimiser ended on non-finite parameters (status "
                          f"{status.value}); the alignment falls back to its starting point")
            return np.array(p0, dtype=np.float64), float("nan"), nit.value
        return x, fval.value, nit.value

    def reference_fd_objective(self, p):
        """(f, g) as the reference's scipy call sees them: g_i = analytic gradient x the share of the
        forward-difference step that survives the fp32 parameter cast."""
        f, g = self.cost_and_grad(p)
        return f, g * fd_survival(p)


def ensemble_depth(depth, scale_invariant=True, shift_invariant=True, output_uncertainty=False,
                   reduction="median", regularizer_strength=0.02, max_iter=50, tol=1e-6, max_res=1024,
                   return_info=False):
    """depth: CUDA tensor [E,1,H,W].  Returns (depth [1,1,H,W], uncertainty [1,1,H,W] | None)."""
    _check_depth_args(depth, reduction, scale_invariant, shift_invariant)
    E, _, H, W = depth.shape
    HW = H * W
    dev = depth.device
    d = depth.to(torch.float32).contiguous()
    info = {}
    st = None
    if scale_invariant or shift_invariant:
        import scipy.optimize

        d_align = d
        if max_res is not None and max(H, W) > max_res:
            f = min(max_res / W, max_res / H)
            from .util.image_util import InterpolationMode, resize
            d_align = resize(d, (int(H * f), int(W * f)), InterpolationMode.NEAREST_EXACT)
        al = DepthAligner(d_align, scale_invariant, shift_invariant, reduction, regularizer_strength)
        p0 = al.init_param()
        if NATIVE_BFGS and isinstance(al.backend, HipStatsBackend):
            # the optimiser and its objective run natively (csrc/bfgs.hip: scipy's BFGS + line searches restated operation for
            # operation; tests/test_host.py compares iterates and evaluation counts with scipy's): an evaluation is the device
            # pass + ~2 us instead of + 60-75 us of Python
            p, cost, nit = al.minimize_native(p0, tol, max_iter)
        else:
            res = scipy.optimize.minimize(al.reference_fd_objective, p0, jac=True, method="BFGS", tol=tol,
                                          options={"maxiter": max_iter, "disp": False})
            p, cost, nit = res.x, float(res.fun), int(res.nit)
        s, t = al._split(p)
        st = torch.from_numpy(np.concatenate([s, t]).astype(np.float32)).to(dev)
        info = dict(param=p, cost=cost, n_eval=al.n_eval, n_iter=nit, aligner=al, status=getattr(al, "status", None))
    med = torch.empty(HW, dtype=torch.float32, device=dev)
    unc = torch.empty(HW, dtype=torch.float32, device=dev) if output_uncertainty else None
    mm = torch.empty(2 + 2 * E, dtype=torch.float32, device=dev)
    scratch = torch.empty(_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    O.launch(O.ens_depth_median(d.reshape(E, HW), st, med, unc, mm, scratch, E=E, HW=HW,
                                reduction=0 if reduction == "median" else 1,
                                has_shift=scale_invariant and shift_invariant))
    if scale_invariant and shift_invariant:
        shift_inv = True
    elif scale_invariant:
        shift_inv = False
    else:
        raise ValueError("Unrecognized alignment.")  # the reference raises here too (:189-190)
    O.launch(O.ens_depth_norm(med, unc, mm, HW=HW, shift_invariant=shift_inv))
    out = med.reshape(1, 1, H, W).to(depth.dtype)
    unc_out = None if unc is None else unc.reshape(1, 1, H, W).to(depth.dtype)
    if return_info:
        return out, unc_out, info
    return out, unc_out


def ensemble_normals(normals, output_uncertainty=False, reduction="closest"):
    """normals: CUDA tensor [E,3,H,W].  Returns ([1,3,H,W], [1,1,H,W] | None)."""
    if normals.dim() != 4 or normals.shape[1] != 3:
        raise ValueError(f"Expecting 4D tensor of shape [B,3,H,W]; got {normals.shape}.")
    if reduction not in ("closest", "mean"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    E, _, H, W = normals.shape   # any ensemble size: the kernel loops over the members
    n = normals.to(torch.float32).contiguous()
    out = torch.empty(3, H * W, dtype=torch.float32, device=n.device)
    unc = torch.empty(H * W, dtype=torch.float32, device=n.device) if output_uncertainty else None
    O.launch(O.ens_normals(n, out, unc, E=E, HW=H * W, reduction=0 if reduction == "closest" else 1))
    return (out.reshape(1, 3, H, W).to(normals.dtype),
            None if unc is None else unc.reshape(1, 1, H, W).to(normals.dtype))


def ensemble_iid(targets, output_uncertainty=False, reduction="median"):
    """targets: CUDA tensor [E,C,H,W] -> ([1,C,H,W], [1,C,H,W] | None): per-element median (+ median
    absolute deviation) or mean (+ unbiased std) over the members (reference ensemble.py:252-270), as one
    pass of the fused median kernel without alignment."""
    if reduction not in ("median", "mean"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    E = targets.shape[0]
    n = targets[0].numel()
    t = targets.to(torch.float32).contiguous()
    dev = t.device
    pred = torch.empty(n, dtype=torch.float32, device=dev)
    unc = torch.empty(n, dtype=torch.float32, device=dev) if output_uncertainty else None
    mm = torch.empty(2 + 2 * E, dtype=torch.float32, device=dev)
    scratch = torch.empty(_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    O.launch(O.ens_depth_median(t.reshape(E, n), None, pred, unc, mm, scratch, E=E, HW=n,
                                reduction=0 if reduction == "median" else 1, has_shift=False))
    shape = (1,) + tuple(targets.shape[1:])
    return pred.reshape(shape).to(targets.dtype), None if unc is None else unc.reshape(shape).to(targets.dtype)
