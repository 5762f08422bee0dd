"""Oracle: test-time ensembling (depth: affine alignment + median; normals: closest-to-mean).

TEST INFRASTRUCTURE - see oracle/__init__.py.  PINNED: ``tests/golden/ensemble_*.npz``
hold outputs of the reference's own /root/reference/marigold/util/ensemble.py executed in
the build container (``oracle/make_golden.py``); ``tests/test_oracle_ensemble.py`` checks
this restatement against them.

Follows /root/reference/marigold/util/ensemble.py:
  ensemble_depth   :39-196  (validation :84-89, init_param :91-105, align :107-118,
                             ensemble :120-136, cost_fn :138-152, compute_param :154-173,
                             final normalisation :184-194)
  ensemble_normals :199-249
  ensemble_iid     :252-270
Semantics kept on purpose: ``torch.median`` picks the LOWER middle element for even E;
the optimiser runs on an fp32 copy while the final affine is applied in the input dtype;
BFGS has no analytic gradient (scipy finite differences), tol=1e-6, maxiter=50.
"""
import numpy as np
import torch


def _check_depth_args(depth, reduction, scale_invariant, shift_invariant):
    if depth.dim() != 4 or depth.shape[1] != 1:
        raise ValueError(f"Expecting 4D tensor of shape [B,1,H,W]; got {depth.shape}.")
    if reduction not in ("mean", "median"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    if not scale_invariant and shift_invariant:
        raise ValueError("Pure shift-invariant ensembling is not supported.")


def depth_init_param(depth32, scale_invariant, shift_invariant):
    E = depth32.shape[0]
    flat = depth32.reshape(E, -1)
    lo, hi = flat.min(dim=1).values, flat.max(dim=1).values
    if scale_invariant and shift_invariant:
        s = 1.0 / (hi - lo).clamp(min=1e-6)
        p = torch.cat((s, -s * lo))
    elif scale_invariant:
        p = 1.0 / hi.clamp(min=1e-6)
    else:
        raise ValueError("Unrecognized alignment.")
    return p.cpu().numpy().astype(np.float64)


def depth_align(depth, param, scale_invariant, shift_invariant):
    E = depth.shape[0]
    if scale_invariant and shift_invariant:
        s, t = np.split(param, 2)
        s = torch.from_numpy(s).to(depth).view(E, 1, 1, 1)
        t = torch.from_numpy(t).to(depth).view(E, 1, 1, 1)
        return depth * s + t
    if scale_invariant:
        s = torch.from_numpy(param).to(depth).view(E, 1, 1, 1)
        return depth * s
    raise ValueError("Unrecognized alignment.")


def depth_reduce(aligned, reduction, want_uncertainty):
    unc = None
    if reduction == "mean":
        pred = aligned.mean(dim=0, keepdim=True)
        if want_uncertainty:
            unc = aligned.std(dim=0, keepdim=True)
    else:
        pred = torch.median(aligned, dim=0, keepdim=True).values
        if want_uncertainty:
            unc = torch.median((aligned - pred).abs(), dim=0, keepdim=True).values
    return pred, unc


def depth_cost(param, depth32, scale_invariant, shift_invariant, reduction, regularizer_strength):
    a = depth_align(depth32, param, scale_invariant, shift_invariant)
    E = a.shape[0]
    cost = 0.0
    for i in range(E):
        for j in range(i + 1, E):
            cost += ((a[i] - a[j]) ** 2).mean().sqrt().item()
    if regularizer_strength > 0:
        pred, _ = depth_reduce(a, reduction, False)
        cost += ((0.0 - pred.min()).abs().item() + (1.0 - pred.max()).abs().item()) \
            * regularizer_strength
    return cost


def ensemble_depth(depth, scale_invariant=True, shift_invariant=True, output_uncertainty=False,
                   reduction="median", regularizer_strength=0.02, max_iter=50, tol=1e-6,
                   max_res=1024, return_param=False):
    _check_depth_args(depth, reduction, scale_invariant, shift_invariant)
    param = None
    if scale_invariant or shift_invariant:
        import scipy.optimize

        d32 = depth.to(torch.float32)
        if max_res is not None and max(d32.shape[2:]) > max_res:
            h, w = d32.shape[2:]
            f = min(max_res / w, max_res / h)
            d32 = torch.nn.functional.interpolate(d32, (int(h * f), int(w * f)), mode="nearest-exact")
        p0 = depth_init_param(d32, scale_invariant, shift_invariant)
        res = scipy.optimize.minimize(
            lambda p: depth_cost(p, d32, scale_invariant, shift_invariant, reduction,
                                 regularizer_strength),
            p0, method="BFGS", tol=tol, options={"maxiter": max_iter, "disp": False})
        param = res.x
        depth = depth_align(depth, param, scale_invariant, shift_invariant)
    depth, unc = depth_reduce(depth, reduction, output_uncertainty)
    hi = depth.max()
    if scale_invariant and shift_invariant:
        lo = depth.min()
    elif scale_invariant:
        lo = 0
    else:
        raise ValueError("Unrecognized alignment.")
    rng = (hi - lo).clamp(min=1e-6)
    depth = (depth - lo) / rng
    if output_uncertainty:
        unc = unc / rng
    if return_param:
        return depth, unc, param
    return depth, unc


def ensemble_normals(normals, output_uncertainty=False, reduction="closest"):
    if normals.dim() != 4 or normals.shape[1] != 3:
        raise ValueError(f"Expecting 4D tensor of shape [B,3,H,W]; got {normals.shape}.")
    if reduction not in ("closest", "mean"):
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    mean = normals.mean(dim=0, keepdim=True)
    mean = mean / torch.norm(mean, dim=1, keepdim=True).clamp(min=1e-6)
    cos = None
    if output_uncertainty or reduction != "mean":
        cos = (mean * normals).sum(dim=1, keepdim=True).clamp(-1, 1)
    unc = None
    if output_uncertainty:
        unc = cos.arccos().mean(dim=0, keepdim=True) / np.pi
    if reduction == "mean":
        return mean, unc
    idx = cos.argmax(dim=0, keepdim=True).repeat(1, 3, 1, 1)
    return torch.gather(normals, 0, idx), unc


def ensemble_iid(targets, output_uncertainty=False, reduction="median"):
    """Per-element median (+MAD) or mean (+std) over the members of [E,C,H,W]."""
    unc = None
    if reduction == "mean":
        pred = targets.mean(dim=0, keepdim=True)
        if output_uncertainty:
            unc = targets.std(dim=0, keepdim=True)
    elif reduction == "median":
        pred = torch.median(targets, dim=0, keepdim=True).values
        if output_uncertainty:
            unc = torch.median((targets - pred).abs(), dim=0, keepdim=True).values
    else:
        raise ValueError(f"Unrecognized reduction method: {reduction}.")
    return pred, unc
