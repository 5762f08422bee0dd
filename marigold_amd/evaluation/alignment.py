"""Affine-invariant evaluation alignment (src/util/alignment.py): the least-squares scale and shift
that map a prediction onto the ground truth over the valid pixels, and depth <-> disparity.

The reference hands the [N,2] system to ``np.linalg.lstsq``; the same minimiser is the solution of the
2x2 normal equations, formed here with fp64 sums (one pass over the valid pixels, no [N,2] matrix).
"""
import numpy as np


def _nearest_downscale(a, factor):
    """What the reference's ``torch.nn.Upsample(scale_factor=factor, mode="nearest")`` does to the ``[1,H,W]``
    tensors it is handed (alignment.py:52-62): a 3-D input is (batch, channels, length), so ONLY THE WIDTH is
    sub-sampled - every row is kept.  Output width floor(W*factor), source column floor(dst * fp32(1/factor))."""
    w = a.shape[-1]
    ow = int(np.floor(w * factor))
    inv = np.float32(1.0 / factor)   # the fp32 source-index arithmetic of the nearest kernel
    ix = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * inv).astype(np.int64), w - 1)
    return a[..., ix]


def align_depth_least_square(gt_arr, pred_arr, valid_mask_arr, return_scale_shift=True, max_resolution=None):
    """-> ``pred_arr * scale + shift`` (full resolution, input shape) [, scale, shift] (alignment.py:35-82)."""
    ori_shape = pred_arr.shape
    gt, pred, valid = np.squeeze(gt_arr), np.squeeze(pred_arr), np.squeeze(valid_mask_arr).astype(bool)
    if max_resolution is not None:
        factor = float(np.min(max_resolution / np.array(ori_shape[-2:])))
        if factor < 1:
            gt, pred, valid = (_nearest_downscale(a, factor) for a in (gt, pred, valid))
    assert gt.shape == pred.shape == valid.shape, f"{gt.shape}, {pred.shape}, {valid.shape}"
    x = pred[valid].astype(np.float64)
    y = gt[valid].astype(np.float64)
    n = x.size
    sx, sy, sxx, sxy = x.sum(), y.sum(), x @ x, x @ y
    det = n * sxx - sx * sx
    scale = (n * sxy - sx * sy) / det
    shift = (sy - scale * sx) / n
    dt = np.result_type(pred_arr.dtype, gt_arr.dtype) if np.issubdtype(pred_arr.dtype, np.floating) else np.float64
    scale, shift = np.asarray([scale], dtype=dt), np.asarray([shift], dtype=dt)
    aligned = (pred_arr * scale + shift).reshape(ori_shape)
    if return_scale_shift:
        return aligned, scale, shift
    return aligned


def depth2disparity(depth, return_mask=False):
    """1/depth where depth > 0, else 0 (alignment.py:86-98)."""
    depth = np.asarray(depth)
    positive = depth > 0
    disparity = np.zeros_like(depth)
    disparity[positive] = 1.0 / depth[positive]
    return (disparity, positive) if return_mask else disparity


def disparity2depth(disparity, **kwargs):
    return depth2disparity(disparity, **kwargs)
