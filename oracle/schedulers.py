"""Oracle: diffusers ``DDIMScheduler`` / ``LCMScheduler`` as the reference drives them.

TEST INFRASTRUCTURE - see oracle/__init__.py.  diffusers is un-vendored; these follow
its published algorithm.  Reference call sites:
/root/reference/marigold/marigold_depth_pipeline.py:423-424 (``set_timesteps`` /
``timesteps``), :466-468 (``step(...).prev_sample``), :348-379 (config fields read by
``_check_inference_step``); scheduler config derivation:
/root/reference/src/trainer/marigold_depth_trainer.py:119-142.
"""
from types import SimpleNamespace

import numpy as np
import torch


def _betas_scaled_linear(beta_start, beta_end, n):
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2


def _rescale_zero_terminal_snr(betas):
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - aT) * (a0 / (a0 - aT))
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1.0 - alphas


class _Base:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 prediction_type="v_prediction", timestep_spacing="trailing",
                 rescale_betas_zero_snr=True, set_alpha_to_one=False, steps_offset=1, **extra):
        betas = _betas_scaled_linear(beta_start, beta_end, num_train_timesteps)
        if rescale_betas_zero_snr:
            betas = _rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = (torch.tensor(1.0) if set_alpha_to_one
                                    else self.alphas_cumprod[0])
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            prediction_type=prediction_type, timestep_spacing=timestep_spacing,
            rescale_betas_zero_snr=rescale_betas_zero_snr, set_alpha_to_one=set_alpha_to_one,
            steps_offset=steps_offset, **extra)
        self.timesteps = None
        self.num_inference_steps = None

    def _x0_eps(self, model_output, sample, a_t):
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(pt)
        return x0, eps


class DDIMScheduler(_Base):
    """eta = 0, clip_sample = False, no thresholding (the Marigold checkpoints' config)."""

    def set_timesteps(self, n, device=None):
        N = self.config.num_train_timesteps
        self.num_inference_steps = n
        sp = self.config.timestep_spacing
        if sp == "leading":
            ts = (np.arange(0, n) * (N // n)).round()[::-1].copy().astype(np.int64)
            ts = ts + self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(N, 0, -N / n)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, N - 1, n).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, generator=None):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0, eps = self._x0_eps(model_output, sample, a_t)
        prev_sample = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=x0)


class LCMScheduler(_Base):
    def __init__(self, original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5,
                 rescale_betas_zero_snr=False, timestep_spacing="leading", **kw):
        super().__init__(rescale_betas_zero_snr=rescale_betas_zero_snr,
                         timestep_spacing=timestep_spacing,
                         original_inference_steps=original_inference_steps,
                         timestep_scaling=timestep_scaling, sigma_data=sigma_data, **kw)
        self._step_index = None

    def set_timesteps(self, n, device=None):
        N = self.config.num_train_timesteps
        orig = self.config.original_inference_steps
        self.num_inference_steps = n
        k = N // orig
        origin = (np.arange(1, orig + 1) * k - 1)[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
        self.timesteps = torch.from_numpy(origin[idx].astype(np.int64))
        self._step_index = None

    def step(self, model_output, timestep, sample, generator=None):
        t = int(timestep)
        if self._step_index is None:
            self._step_index = int((self.timesteps == t).nonzero()[0])
        i = self._step_index
        t_prev = int(self.timesteps[i + 1]) if i + 1 < len(self.timesteps) else t
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.final_alpha_cumprod
        s = t * self.config.timestep_scaling
        sd = self.config.sigma_data
        c_skip = sd ** 2 / (s ** 2 + sd ** 2)
        c_out = s / (s ** 2 + sd ** 2) ** 0.5
        x0, _ = self._x0_eps(model_output, sample, a_t)
        denoised = c_out * x0 + c_skip * sample
        if i != self.num_inference_steps - 1:
            noise = torch.randn(model_output.shape, generator=generator,
                                dtype=denoised.dtype, device=denoised.device)
            prev_sample = a_prev ** 0.5 * denoised + (1 - a_prev) ** 0.5 * noise
        else:
            prev_sample = denoised
        self._step_index += 1
        return SimpleNamespace(prev_sample=prev_sample, denoised=denoised)
