"""Property tests (hypothesis) of the host logic around the hot path: scheduler update algebra, evaluation
alignment / metrics invariances, prediction-file naming.  CPU only, small examples."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from marigold_amd import evaluation as E
from marigold_amd.evaluation import datasets as D, metrics as M
from marigold_amd.schedulers import DDIMScheduler, LCMScheduler

SET = dict(max_examples=25, deadline=None)


@settings(**SET)
@given(n=st.integers(1, 50), spacing=st.sampled_from(["trailing", "leading"]), zero_snr=st.booleans(),
       pred=st.sampled_from(["v_prediction", "epsilon", "sample"]), seed=st.integers(0, 10))
def test_ddim_step_moves_a_clean_noise_mixture_along_its_own_trajectory(n, spacing, zero_snr, pred, seed):
    """If the model output is the TRUE v / epsilon / x0 of x_t = sqrt(a) x0 + sqrt(1-a) eps, one DDIM step (eta = 0)
    must land exactly on sqrt(a') x0 + sqrt(1-a') eps."""
    s = DDIMScheduler(timestep_spacing=spacing, rescale_betas_zero_snr=zero_snr, prediction_type=pred)
    s.set_timesteps(n)
    r = np.random.default_rng(seed)
    x0, eps = r.normal(size=16), r.normal(size=16)
    for i in range(n):
        t = int(s.timesteps[i])
        a = float(s.alphas_cumprod[t])
        if (pred == "epsilon" and a < 1e-12) or (pred == "sample" and 1 - a < 1e-12):
            continue                                   # the parametrisation itself is singular there
        prev = t - 1000 // n
        ap = float(s.alphas_cumprod[prev]) if prev >= 0 else s.final_alpha_cumprod
        xt = a ** 0.5 * x0 + (1 - a) ** 0.5 * eps
        out = {"v_prediction": a ** 0.5 * eps - (1 - a) ** 0.5 * x0, "epsilon": eps, "sample": x0}[pred]
        cx, cm, cn = s.step_coefficients(i)
        assert cn == 0.0
        np.testing.assert_allclose(cx * xt + cm * out, ap ** 0.5 * x0 + (1 - ap) ** 0.5 * eps, atol=2e-6 / max(a, 1e-3) ** 0.5)


@settings(**SET)
@given(n=st.integers(1, 50))
def test_lcm_timesteps_and_boundary_condition(n):
    """LCM: strictly decreasing timesteps from the 50-step grid starting at 999; the last step returns the denoised
    sample (no re-noising), c_skip -> 0 and c_out -> 1 for large t."""
    s = LCMScheduler()
    s.set_timesteps(n)
    ts = s.timesteps.numpy()
    assert ts[0] == 999 and (np.diff(ts) < 0).all() and ((ts + 1) % 20 == 0).all()
    cx, cm, cn = s.step_coefficients(n - 1)
    assert cn == 0.0 and not s.needs_noise(n - 1)
    if n > 1:
        assert s.needs_noise(0) and s.step_coefficients(0)[2] > 0
    a = float(s.alphas_cumprod[int(ts[-1])])
    np.testing.assert_allclose([cx, cm], [a ** 0.5, -(1 - a) ** 0.5], rtol=1e-4)   # ~ the plain x0 prediction


@settings(**SET)
@given(scale=st.floats(0.05, 20), shift=st.floats(-5, 5), seed=st.integers(0, 100), h=st.integers(4, 24), w=st.integers(4, 24))
def test_least_squares_alignment_recovers_an_affine_map(scale, shift, seed, h, w):
    r = np.random.default_rng(seed)
    gt = r.uniform(0.5, 10, (h, w))
    mask = r.uniform(size=(h, w)) > 0.3
    mask[0, :2] = True
    gt[0, 1] = gt[0, 0] + 1.0                           # at least two distinct valid values
    pred = (gt - shift) / scale
    aligned, s, t = E.align_depth_least_square(gt, pred, mask, True)
    np.testing.assert_allclose([float(s[0]), float(t[0])], [scale, shift], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(aligned, gt, rtol=1e-7, atol=1e-7)
    vals = {n: getattr(M, n)(aligned.astype(np.float32), gt.astype(np.float32), mask) for n in M.DEPTH_METRICS}
    assert vals["abs_relative_difference"] < 1e-6 and vals["delta1_acc"] == 1.0 and vals["rmse_linear"] < 1e-5


@settings(**SET)
@given(seed=st.integers(0, 100), k=st.floats(0.1, 10))
def test_depth_metrics_scale_behaviour(seed, k):
    """Relative metrics (abs-rel, delta, log-RMSE, SILog) do not change when prediction and ground truth are scaled
    together; rmse_linear scales with them; silog additionally ignores a scale on the prediction alone."""
    r = np.random.default_rng(seed)
    gt = r.uniform(0.5, 10, (12, 17)).astype(np.float32)
    pred = (gt * r.uniform(0.7, 1.4, gt.shape)).astype(np.float32)
    mask = r.uniform(size=gt.shape) > 0.2
    kk = np.float32(k)
    for name in ("abs_relative_difference", "delta1_acc", "delta2_acc", "rmse_log", "log10", "silog_rmse"):
        np.testing.assert_allclose(getattr(M, name)(pred * kk, gt * kk, mask), getattr(M, name)(pred, gt, mask),
                                   rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(M.rmse_linear(pred * kk, gt * kk, mask), k * M.rmse_linear(pred, gt, mask), rtol=2e-4)
    np.testing.assert_allclose(M.silog_rmse(pred * kk, gt, mask), M.silog_rmse(pred, gt, mask), rtol=1e-3, atol=2e-3)


@settings(**SET)
@given(stem=st.text(alphabet="abcdefgh0123456789", min_size=1, max_size=8), ext=st.sampled_from([".png", ".jpg", ".JPG"]))
def test_prediction_names(stem, ext):
    assert E.get_pred_name(stem + ext, D.PredNameMode.id, ".npy") == "pred_" + stem + ".npy"
    assert E.get_pred_name("rgb_" + stem + ext, D.PredNameMode.rgb_id, ".npy") == "pred_" + stem + ".npy"
    assert E.get_pred_name(stem + "_rgb" + ext, D.PredNameMode.i_d_rgb, ".npy") == stem + "_pred.npy"
    assert E.get_pred_name("rgb_cam_00_" + stem + ext, D.PredNameMode.rgb_i_d, ".npy") == "pred_cam_00_" + stem + ".npy"


def test_normals_angular_error_of_rotated_field():
    r = np.random.default_rng(0)
    n = r.normal(size=(3, 10, 12)).astype(np.float32)
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    th = np.deg2rad(11.0)
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    err = M.compute_cosine_error(np.einsum("ij,jhw->ihw", rot, n), n, masked=True)
    # a rotation about z by 11 deg moves a unit vector by 11 deg * sin(polar angle): never more than 11
    assert err.max() <= 11.0 + 1e-2 and M.sub11_25_error(err) == 100.0 and M.sub5_error(err) < 100.0
    assert pytest.approx(M.rmse_angular_error(err) ** 2, rel=1e-3) == float(np.mean(err.astype(np.float64) ** 2))
