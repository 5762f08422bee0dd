"""Seeded synthetic checkpoints and inputs (no network, no weights on disk - BASELINE.md §4).

Weights follow the diffusers key layout of ``arch.py`` so a real checkpoint is a drop-in.
Each tensor is drawn from its own CPU generator seeded by crc32(key) ^ seed, so the values do
not depend on enumeration order.  Conv/Linear weights ~ N(0, gain/fan_in); norm affine
parameters are non-trivial (gamma = 1 + 0.1 n, beta = 0.1 n) so the affine paths are tested.
Inputs follow SURVEY.md §8(d): smooth low-frequency RGB image, fixed text context, initial
latents drawn from a CPU generator.
"""
import zlib

import torch

from .arch import UNetConfig, VAEConfig, unet_param_shapes, vae_param_shapes


def _draw(key, shape, seed):
    g = torch.Generator("cpu").manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 2)
    is_norm = any(s in key for s in ("norm", "group_norm"))
    if key.endswith(".weight") and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
    if key.endswith(".weight") and is_norm:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if key.endswith(".bias") and is_norm:
        return 0.1 * torch.randn(shape, generator=g)
    del leaf
    return 0.05 * torch.randn(shape, generator=g)  # conv / linear bias


def synthetic_state_dict(shapes, seed):
    return {k: _draw(k, tuple(s), seed) for k, s in shapes.items()}


_STREAM_WRITERS = ("proj_in.weight", "attn1.to_out.0.weight", "attn2.to_out.0.weight", "ff.net.2.weight")


def plant_heavy_tails(sd, seed=1234, outlier_frac=0.02, outlier_gain=30.0, affine_gain=8.0, gate_gain=4.0):
    """SD-like activation statistics for the parity stress run (in place, deterministic per key): the layers
    that WRITE the transformer residual stream get 2 % of their output channels at ``outlier_gain`` x (outlier channels,
    |mean| / std >> 1 on the rows the folded LayerNorm cancels), every LayerNorm / GroupNorm gamma and beta a few entries
    at ``affine_gain`` x, and the GEGLU gate rows ``gate_gain`` x (pre-activations beyond the polynomial CDF's +-4)."""
    for key, t in sd.items():
        g = torch.Generator("cpu").manual_seed((zlib.crc32(key.encode()) ^ (seed * 40503) ^ 0x5BD1E995) & 0x7FFFFFFF)
        if key.endswith(_STREAM_WRITERS):
            n = max(2, int(round(outlier_frac * t.shape[0])))
            t[torch.randperm(t.shape[0], generator=g)[:n]] *= outlier_gain
        elif ("norm" in key) and t.dim() == 1:
            t[torch.randperm(t.shape[0], generator=g)[:4]] *= affine_gain
        elif key.endswith("ff.net.0.proj.weight"):
            t[t.shape[0] // 2:] *= gate_gain
    return sd


def synthetic_unet_state_dict(cfg: UNetConfig = UNetConfig(), seed: int = 1234, heavy_tail: bool = False):
    sd = synthetic_state_dict(unet_param_shapes(cfg), seed)
    return plant_heavy_tails(sd, seed) if heavy_tail else sd


def synthetic_vae_state_dict(cfg: VAEConfig = VAEConfig(), seed: int = 1234):
    return synthetic_state_dict(vae_param_shapes(cfg), seed + 1)


def synthetic_text_embedding(cross_dim: int = 1024, seed: int = 7):
    """Stands in for CLIP("") with padding="do_not_pad": 2 tokens
    (/root/reference/marigold/marigold_depth_pipeline.py:381-394)."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(1, 2, cross_dim, generator=g)


def synthetic_image(height: int = 768, width: int = 768, seed: int = 0):
    """uint8 [1,3,H,W]: low-frequency content (random 24x24 grid, bilinear up-sampled)."""
    g = torch.Generator("cpu").manual_seed(seed)
    coarse = torch.rand(1, 3, 24, 24, generator=g)
    img = torch.nn.functional.interpolate(coarse, size=(height, width), mode="bilinear",
                                          align_corners=False)
    return (img * 255.0).round().clamp(0, 255).to(torch.uint8)


def synthetic_latents(n, h, w, seed: int = 2024, dtype=torch.float32):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(n, 4, h, w, generator=g, dtype=dtype)


def heavy_tailed_rows(M, C, seed=0, outlier_frac=0.02, outlier_gain=(30.0, 100.0), mean_sigma=20.0):
    """Activation rows with the statistics real SD-v2 residual streams show and the seeded N(0, 1/fan_in) weights above
    never produce: 1-2 % outlier channels at 30-100x, row means up to ``mean_sigma`` standard
    deviations.  Used by the parity stress tests of the fusions that cancel large numbers (the folded LayerNorm) or
    approximate on a bounded interval (the polynomial GELU)."""
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.randn(M, C, generator=g)
    n_out = max(1, int(round(outlier_frac * C)))
    idx = torch.randperm(C, generator=g)[:n_out]
    gain = outlier_gain[0] + (outlier_gain[1] - outlier_gain[0]) * torch.rand(n_out, generator=g)
    x[:, idx] *= gain
    x += mean_sigma * (torch.rand(M, 1, generator=g) * 2 - 1)
    return x


def heavy_tailed_affine(C, seed=0, n_large=4, gain=8.0):
    """LayerNorm / GroupNorm (gamma, beta) with a few large entries."""
    g = torch.Generator("cpu").manual_seed(seed + 1)
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    idx = torch.randperm(C, generator=g)[:n_large]
    gamma[idx] *= gain
    beta[idx] *= gain
    return gamma, beta
