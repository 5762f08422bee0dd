"""Oracle: the reference's own parity metrics (TEST INFRASTRUCTURE - see oracle/__init__.py).

Affine-invariant depth:  /root/reference/src/util/alignment.py:35-82 (least-squares
scale/shift of prediction onto target) then /root/reference/src/util/metric.py:64-74
(abs-rel), :91-103 (rmse), :148-149 (delta1).
Normals: /root/reference/src/util/metric.py:194-223 (per-pixel angular error in degrees).
"""
import numpy as np
import torch


def align_least_squares(target, pred):
    """Return (aligned_pred, scale, shift) minimising ||scale*pred + shift - target||."""
    t = np.asarray(target, dtype=np.float64).reshape(-1, 1)
    p = np.asarray(pred, dtype=np.float64).reshape(-1, 1)
    A = np.concatenate([p, np.ones_like(p)], axis=-1)
    (scale,), (shift,) = np.linalg.lstsq(A, t, rcond=None)[0]
    return np.asarray(pred, dtype=np.float64) * scale + shift, float(scale), float(shift)


def affine_invariant_depth_errors(target, pred, floor=1e-3):
    """abs-rel / rmse / delta1 after LS alignment.  ``floor`` guards the division where the
    (synthetic-weight) target depth is ~0; the reference evaluates on metric GT > 0."""
    tgt = np.asarray(target, dtype=np.float64)
    al, s, b = align_least_squares(tgt, pred)
    m = tgt > floor
    absrel = float(np.mean(np.abs(al[m] - tgt[m]) / tgt[m])) if m.any() else 0.0
    rmse = float(np.sqrt(np.mean((al - tgt) ** 2)))
    ratio = np.maximum(al[m] / tgt[m], tgt[m] / np.maximum(al[m], 1e-12))
    d1 = float(np.mean(ratio < 1.25)) if m.any() else 1.0
    return dict(abs_rel=absrel, rmse=rmse, delta1=d1, scale=s, shift=b)


def angular_error_deg(pred, target):
    """Per-pixel angle between two [3,H,W] normal maps, degrees."""
    p = torch.as_tensor(pred, dtype=torch.float64).reshape(3, -1)
    t = torch.as_tensor(target, dtype=torch.float64).reshape(3, -1)
    c = torch.cosine_similarity(p, t, dim=0).clamp(-1.0, 1.0)
    return (torch.acos(c) * 180.0 / np.pi).numpy()
