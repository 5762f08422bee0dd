"""Evaluation metrics of the reference's benchmark protocol (src/util/metric.py), restated on numpy.

Predictions are read back from ``.npy`` files and compared with ground truth one image at a time:
the work is a handful of masked reductions per image, dominated by file IO, so it stays on the host
(fp32 element arithmetic like the reference's torch-fp32 tensors, fp64 accumulation for the sums -
results agree with the reference's to ~1e-6 relative, see tests/test_evaluation.py).

Depth metrics take ``(pred, gt, valid_mask)`` as ``[H,W]`` arrays (src/util/metric.py:64-199);
normals metrics take the flat per-pixel angular error in degrees (:206-279); the IID helpers follow
:285-375.
"""
import numpy as np

# ---- depth ---------------------------------------------------------------------------------------


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def _masked_mean(values, valid_mask):
    """sum over valid pixels / number of valid pixels (metric.py: entries outside the mask are zeroed,
    n = mask.sum())."""
    if valid_mask is None:
        return float(values.sum(dtype=np.float64) / values.size)
    m = np.asarray(valid_mask, dtype=bool)
    return float(values[m].sum(dtype=np.float64) / m.sum())


def abs_relative_difference(output, target, valid_mask=None):
    o, t = _f32(output), _f32(target)
    with np.errstate(divide="ignore", invalid="ignore"):
        return _masked_mean(np.abs(o - t) / t, valid_mask)


def squared_relative_difference(output, target, valid_mask=None):
    o, t = _f32(output), _f32(target)
    with np.errstate(divide="ignore", invalid="ignore"):
        return _masked_mean(np.abs(o - t) ** 2 / t, valid_mask)


def rmse_linear(output, target, valid_mask=None):
    d = _f32(output) - _f32(target)
    return float(np.sqrt(_masked_mean(d * d, valid_mask)))


def rmse_log(output, target, valid_mask=None):
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.log(_f32(output)) - np.log(_f32(target))
    return float(np.sqrt(_masked_mean(d * d, valid_mask)))


def log10(output, target, valid_mask=None):
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.abs(np.log10(_f32(output)) - np.log10(_f32(target)))
    return _masked_mean(d, valid_mask)


def threshold_percentage(output, target, threshold_val, valid_mask=None):
    o, t = _f32(output), _f32(target)
    with np.errstate(divide="ignore", invalid="ignore"):
        worst = np.maximum(o / t, t / o)
    return _masked_mean((worst < threshold_val).astype(np.float32), valid_mask)


def delta1_acc(pred, gt, valid_mask=None):
    return threshold_percentage(pred, gt, 1.25, valid_mask)


def delta2_acc(pred, gt, valid_mask=None):
    return threshold_percentage(pred, gt, 1.25 ** 2, valid_mask)


def delta3_acc(pred, gt, valid_mask=None):
    return threshold_percentage(pred, gt, 1.25 ** 3, valid_mask)


def i_rmse(output, target, valid_mask=None):
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.float32(1.0) / _f32(output) - np.float32(1.0) / _f32(target)
    return float(np.sqrt(_masked_mean(d * d, valid_mask)))


def silog_rmse(depth_pred, depth_gt, valid_mask=None):
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.log(_f32(depth_pred)) - np.log(_f32(depth_gt))
    first = _masked_mean(d * d, valid_mask)
    second = _masked_mean(d, valid_mask) ** 2
    return float(np.sqrt(first - second) * 100.0)


DEPTH_METRICS = ("abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10",
                 "delta1_acc", "delta2_acc", "delta3_acc", "i_rmse", "silog_rmse")   # script/depth/eval.py:58-69

# ---- surface normals -----------------------------------------------------------------------------


def compute_cosine_error(pred_norm, gt_norm, masked=False):
    """Per-pixel angle in degrees between ``[3,H,W]`` (or ``[1,3,H,W]``) normal maps, flattened; with
    ``masked`` the pixels whose ground-truth vector is zero are dropped (metric.py:206-233).  Like
    ``torch.cosine_similarity`` each norm is clamped below by 1e-8."""
    p, g = _f32(pred_norm), _f32(gt_norm)
    if p.ndim == 4:
        p = p[0]
    if g.ndim == 4:
        g = g[0]
    assert p.shape[0] == 3 and g.shape[0] == 3, "Channel dim should be the first dimension!"
    p, g = p.reshape(3, -1), g.reshape(3, -1)
    if masked:
        keep = np.sqrt((g * g).sum(0)) > 0
        p, g = p[:, keep], g[:, keep]
    eps = np.float32(1e-8)
    pn = np.maximum(np.sqrt((p * p).sum(0)), eps)
    gn = np.maximum(np.sqrt((g * g).sum(0)), eps)
    cos = np.clip(((p / pn) * (g / gn)).sum(0), -1.0, 1.0)
    return (np.arccos(cos) * np.float32(180.0 / np.pi)).astype(np.float32)


def mean_angular_error(cosine_error):
    return round(float(np.average(cosine_error)), 4)


def median_angular_error(cosine_error):
    return round(float(np.median(cosine_error)), 4)


def rmse_angular_error(cosine_error):
    return round(float(np.sqrt(np.sum(cosine_error * cosine_error) / cosine_error.shape[0])), 4)


def _sub(cosine_error, deg):
    return round(100.0 * float(np.sum(cosine_error < deg) / cosine_error.shape[0]), 4)


def sub5_error(cosine_error):
    return _sub(cosine_error, 5)


def sub7_5_error(cosine_error):
    return _sub(cosine_error, 7.5)


def sub11_25_error(cosine_error):
    return _sub(cosine_error, 11.25)


def sub22_5_error(cosine_error):
    return _sub(cosine_error, 22.5)


def sub30_error(cosine_error):
    return _sub(cosine_error, 30)


NORMALS_METRICS = ("mean_angular_error", "median_angular_error", "sub5_error", "sub7_5_error", "sub11_25_error",
                   "sub22_5_error", "sub30_error")   # script/normals/eval.py:47-55

# ---- intrinsic image decomposition ---------------------------------------------------------------


def compute_alignment_scale(pred, gt, valid_mask=None):
    """Least-squares scalar s minimising |s*pred - gt|^2 over (valid) pixels (metric.py:319-334)."""
    p, g = _f32(pred).squeeze(), _f32(gt).squeeze()
    assert p.shape[0] == 3 and g.shape[0] == 3, "First dim should be channel dim"
    if valid_mask is not None:
        m = np.asarray(valid_mask, dtype=bool).squeeze()
        p, g = p[m], g[m]
    p64, g64 = p.astype(np.float64).ravel(), g.astype(np.float64).ravel()
    return float(p64 @ g64 / (p64 @ p64))


def quantile_map(pred, gt, valid_mask=None, percentile=90, desired=0.8):
    """Scale both images so that the 90th percentile of the ground-truth brightness sits at 0.8, clamp to
    [0,1] (metric.py:337-375)."""
    p, g = _f32(pred).squeeze(), _f32(gt).squeeze()
    assert g.shape[0] == 3, "channel dim must be first dim"
    brightness = np.float32(0.3) * g[0] + np.float32(0.59) * g[1] + np.float32(0.11) * g[2]
    if valid_mask is not None:
        brightness = brightness[np.asarray(valid_mask, dtype=bool).squeeze()[0]]
    q = float(np.quantile(brightness.ravel().astype(np.float32), percentile / 100.0))   # linear interpolation
    scale = np.float32(0.0 if q < 1e-4 else desired / q)
    return np.clip(scale * p, 0, 1)[None], np.clip(scale * g, 0, 1)[None]


def psnr(pred, gt, data_range=1.0):
    """10 log10(range^2 / mse) over all given elements (torchmetrics PeakSignalNoiseRatio, elementwise-mean)."""
    d = _f32(pred).astype(np.float64) - _f32(gt).astype(np.float64)
    with np.errstate(divide="ignore"):   # identical images score +inf, like the reference's torchmetrics call
        return float(10.0 * np.log10(data_range ** 2 / np.mean(d * d)))


def ssim(pred, gt, data_range=1.0, sigma=1.5, kernel_size=11, k1=0.01, k2=0.03):
    """Structural similarity of ``[1,C,H,W]`` images with the Gaussian window the reference's
    torchmetrics call uses by default (11x11, sigma 1.5, reflect padding, padded border cropped)."""
    x, y = _f32(pred).astype(np.float64), _f32(gt).astype(np.float64)
    x, y = x.reshape((-1,) + x.shape[-2:]), y.reshape((-1,) + y.shape[-2:])
    r = kernel_size // 2
    ax = np.arange(kernel_size, dtype=np.float64) - r
    k = np.exp(-(ax / sigma) ** 2 / 2)
    k /= k.sum()

    def blur(a):
        a = np.pad(a, ((0, 0), (r, r), (r, r)), mode="reflect")
        a = sum(k[i] * a[:, i:i + a.shape[1] - 2 * r, :] for i in range(kernel_size))
        return sum(k[i] * a[:, :, i:i + a.shape[2] - 2 * r] for i in range(kernel_size))

    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    mx, my = blur(x), blur(y)
    sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    s = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
    return float(s[:, r:-r, r:-r].mean())


def compute_iid_metric(pred, gt, target_name, metric_name, valid_mask=None):
    """One IID metric of one target (metric.py:285-316): shading / residual are up to scale, so they are
    first scale-aligned to the ground truth and brightness-mapped to [0,1]; PSNR is taken over the valid
    elements, SSIM with the invalid ones zeroed."""
    p, g = _f32(pred), _f32(gt)
    if target_name in ("shading", "residual"):
        p = np.float32(compute_alignment_scale(p, g, valid_mask)) * p
        p, g = quantile_map(p, g, valid_mask)
    p = p[None] if p.ndim == 3 else p
    g = g[None] if g.ndim == 3 else g
    if valid_mask is not None:
        m = np.asarray(valid_mask, dtype=bool)
        m = m[None] if m.ndim == 3 else m
        if metric_name == "psnr":
            return psnr(p[m], g[m])
        p, g = np.where(m, p, 0), np.where(m, g, 0)
    if metric_name == "psnr":
        return psnr(p, g)
    if metric_name == "ssim":
        return ssim(p, g)
    raise NotImplementedError(f"IID metric '{metric_name}' (LPIPS needs pretrained network weights that are not "
                              f"part of this engine)")


# ---- running averages ----------------------------------------------------------------------------


class MetricTracker:
    """Running per-key average (metric.py:29-57), without the pandas / tensorboard plumbing."""

    def __init__(self, *keys):
        self._keys = list(keys)
        self.reset()

    def reset(self):
        self._total = {k: 0.0 for k in self._keys}
        self._count = {k: 0 for k in self._keys}

    def update(self, key, value, n=1):
        self._total[key] += value * n
        self._count[key] += n

    def avg(self, key):
        return self._total[key] / self._count[key] if self._count[key] else 0.0

    def result(self):
        return {k: self.avg(k) for k in self._keys}
