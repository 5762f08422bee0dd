"""DDIM / LCM schedulers for the native denoising loop.

Mirror of the diffusers ``DDIMScheduler`` / ``LCMScheduler`` surface the reference touches
(marigold_depth_pipeline.py:348-379 ``config.timestep_spacing`` / ``rescale_betas_zero_snr`` +
isinstance checks, :423-424 ``set_timesteps`` / ``timesteps``, :466-468 ``step``).  Both updates
are linear in (sample, model_output, noise), so a step is ONE launch of the ``sched_step``
kernel: x <- cx*x + cm*model_out + cn*noise, with (cx, cm, cn) computed here in fp64 and baked
into the step's op (no table lookups or host syncs inside the loop).
"""
from types import SimpleNamespace

import numpy as np
import torch


def _alphas_cumprod(n, beta_start, beta_end, zero_snr):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    if zero_snr:
        ab = np.sqrt(np.cumprod(1.0 - betas))
        a0, aT = ab[0], ab[-1]
        ab = (ab - aT) * (a0 / (a0 - aT))
        return ab ** 2
    return np.cumprod(1.0 - betas)


class _SchedulerBase:
    _defaults = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                     beta_schedule="scaled_linear", prediction_type="v_prediction",
                     timestep_spacing="trailing", rescale_betas_zero_snr=True,
                     set_alpha_to_one=False, steps_offset=1, clip_sample=False)

    def __init__(self, **kw):
        cfg = dict(self._defaults)
        cfg.update(kw)
        if cfg["beta_schedule"] != "scaled_linear":
            raise ValueError(f"unsupported beta_schedule {cfg['beta_schedule']}")
        if cfg.get("clip_sample"):
            raise ValueError("clip_sample=True is not supported (Marigold checkpoints use False)")
        self.config = SimpleNamespace(**cfg)
        self.alphas_cumprod = _alphas_cumprod(cfg["num_train_timesteps"], cfg["beta_start"],
                                              cfg["beta_end"], cfg["rescale_betas_zero_snr"])
        self.final_alpha_cumprod = 1.0 if cfg["set_alpha_to_one"] else float(self.alphas_cumprod[0])
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, cfg: dict):
        known = {k: v for k, v in cfg.items() if not k.startswith("_")}
        return cls(**known)

    def signature(self):
        return (type(self).__name__, tuple(sorted((k, str(v)) for k, v in vars(self.config).items())))

    def _x0_coeffs(self, a):
        """x0 = kx * x + km * model_output."""
        b = 1.0 - a
        pt = self.config.prediction_type
        if pt == "v_prediction":
            return a ** 0.5, -(b ** 0.5)
        if pt == "epsilon":
            return 1.0 / a ** 0.5, -(b ** 0.5) / a ** 0.5
        if pt == "sample":
            return 0.0, 1.0
        raise ValueError(f"unknown prediction_type {pt}")

    def _eps_coeffs(self, a):
        """eps = ex * x + em * model_output."""
        b = 1.0 - a
        pt = self.config.prediction_type
        if pt == "v_prediction":
            return b ** 0.5, a ** 0.5
        if pt == "epsilon":
            return 0.0, 1.0
        return 1.0 / b ** 0.5, -(a ** 0.5) / b ** 0.5  # sample

    def needs_noise(self, i):
        return False

    def step(self, model_output, timestep, sample, generator=None):
        """Step-wise mirror of diffusers' ``scheduler.step`` on device tensors (HIP kernel)."""
        from . import ops as O
        i = int((self.timesteps == int(timestep)).nonzero()[0])
        cx, cm, cn = self.step_coefficients(i)
        noise = None
        if self.needs_noise(i):
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                dtype=torch.float32)
        out = torch.empty_like(sample, dtype=torch.float32)
        O.launch(O.sched_step(sample.float().contiguous(), model_output.float().contiguous(), noise, out,
                              n=sample.numel(), cx=cx, cm=cm, cn=cn))
        return SimpleNamespace(prev_sample=out)


class DDIMScheduler(_SchedulerBase):
    """eta = 0 (deterministic), no clipping / thresholding."""

    def set_timesteps(self, n, device=None):
        N = self.config.num_train_timesteps
        sp = self.config.timestep_spacing
        if sp == "leading":
            ts = (np.arange(0, n) * (N // n)).round()[::-1].astype(np.int64) + self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(N, 0, -N / n)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, N - 1, n).round()[::-1].astype(np.int64)
        else:
            raise ValueError(f"unsupported timestep_spacing {sp}")
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts.copy())

    def step_coefficients(self, i):
        t = int(self.timesteps[i])
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[prev]) if prev >= 0 else self.final_alpha_cumprod
        kx, km = self._x0_coeffs(a)
        ex, em = self._eps_coeffs(a)
        sa, sb = ap ** 0.5, (1.0 - ap) ** 0.5
        return sa * kx + sb * ex, sa * km + sb * em, 0.0


class LCMScheduler(_SchedulerBase):
    _defaults = dict(_SchedulerBase._defaults, timestep_spacing="leading", rescale_betas_zero_snr=False,
                     original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5)

    def set_timesteps(self, n, device=None):
        N, orig = self.config.num_train_timesteps, self.config.original_inference_steps
        k = N // orig
        origin = (np.arange(1, orig + 1) * k - 1)[::-1]
        idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(origin[idx].astype(np.int64).copy())

    def needs_noise(self, i):
        return i != self.num_inference_steps - 1

    def step_coefficients(self, i):
        t = int(self.timesteps[i])
        last = i == self.num_inference_steps - 1
        t_prev = t if last else int(self.timesteps[i + 1])
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[t_prev]) if t_prev >= 0 else self.final_alpha_cumprod
        s = t * self.config.timestep_scaling
        sd = self.config.sigma_data
        c_skip = sd ** 2 / (s ** 2 + sd ** 2)
        c_out = s / (s ** 2 + sd ** 2) ** 0.5
        kx, km = self._x0_coeffs(a)
        dx, dm = c_out * kx + c_skip, c_out * km
        if last:
            return dx, dm, 0.0
        return ap ** 0.5 * dx, ap ** 0.5 * dm, (1.0 - ap) ** 0.5
