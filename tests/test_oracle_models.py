"""Structural + golden checks of the oracle's UNet / VAE / scheduler restatements."""
import math
import os

import numpy as np
import pytest
import torch

from marigold_amd import synthetic as syn
from marigold_amd.arch import (TINY_UNET, TINY_VAE, UNetConfig, unet_param_shapes,
                               unet_up_resnet_channels, vae_param_shapes)
from oracle import pipeline as opipe
from oracle.schedulers import DDIMScheduler, LCMScheduler
from oracle.sd2_unet import UNet2DConditionModel
from oracle.sd2_vae import AutoencoderKL


def _count(d, *prefixes):
    return sum(math.prod(s) for k, s in d.items() if k.startswith(prefixes))


def test_param_counts_match_published_sd2():
    u, v = unet_param_shapes(), vae_param_shapes()
    assert round(_count(u, "") / 1e6, 2) == 865.92
    assert round(_count(v, "encoder", "quant_conv") / 1e6, 2) == 34.16
    assert round(_count(v, "decoder", "post_quant_conv") / 1e6, 2) == 49.49


def test_state_dict_keys_match_arch_tables():
    with torch.device("meta"):
        m, a = UNet2DConditionModel(), AutoencoderKL()
    assert {k: tuple(p.shape) for k, p in m.state_dict().items()} == dict(unet_param_shapes())
    assert {k: tuple(p.shape) for k, p in a.state_dict().items()} == dict(vae_param_shapes())


def test_skip_stack_widths():
    rows = unet_up_resnet_channels(UNetConfig())
    assert [r[2] + r[3] for r in rows] == [2560, 2560, 2560, 2560, 2560, 1920,
                                            1920, 1280, 960, 960, 640, 640]


def test_scheduler_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "scheduler_tables.npz"))
    for n in (10, 4, 1):
        s = DDIMScheduler()
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"ddim_trailing_{n}"].tolist()
    for n in (10, 1):
        s = DDIMScheduler(timestep_spacing="leading", rescale_betas_zero_snr=False)
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"ddim_leading_{n}"].tolist()
    for n in (4, 1):
        s = LCMScheduler()
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"lcm_{n}"].tolist()
    z = DDIMScheduler()
    assert float(z.alphas_cumprod[999]) == 0.0          # zero terminal SNR
    assert abs(float(z.alphas_cumprod[0]) - 0.99915) < 1e-4


def test_zero_snr_first_step_is_minus_v():
    s = DDIMScheduler()
    s.set_timesteps(1)
    x, v = torch.randn(1, 4, 4, 4), torch.randn(1, 4, 4, 4)
    out = s.step(v, 999, x)
    torch.testing.assert_close(out.pred_original_sample, -v)


@pytest.fixture(scope="module")
def tiny():
    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(min(16, usable_cores()))
    unet = UNet2DConditionModel(block_out_channels=TINY_UNET.block_out_channels,
                                attention_head_dim=TINY_UNET.heads,
                                cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict(TINY_UNET))
    vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels).eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict(TINY_VAE))
    return unet, vae


def test_tiny_pipeline_golden(tiny, golden_dir):
    unet, vae = tiny
    g = np.load(os.path.join(golden_dir, "tiny_pipeline.npz"))
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    img = syn.synthetic_image(64, 128, seed=0)
    lat0 = syn.synthetic_latents(3, 8, 16, seed=2024)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    with torch.no_grad():
        rl = opipe.encode_rgb(vae, rgb)
        np.testing.assert_allclose(rl.numpy(), g["rgb_latent"], atol=1e-4)
        d = opipe.single_infer("depth", unet, vae, DDIMScheduler(), rgb.expand(3, -1, -1, -1),
                               lat0, ctx, 4)
        np.testing.assert_allclose(d.numpy(), g["depth_ddim4"], atol=1e-3)
        gen = torch.Generator("cpu").manual_seed(99)
        dl = opipe.single_infer("depth", unet, vae, LCMScheduler(), rgb, lat0[:1], ctx, 3,
                                generator=gen)
        np.testing.assert_allclose(dl.numpy(), g["depth_lcm3"], atol=1e-3)


def test_unet_odd_latent_size_uses_upsample_size(tiny):
    unet, _ = tiny
    x = torch.randn(1, 8, 12, 20)
    y = unet(x, torch.tensor(500), torch.randn(1, 2, TINY_UNET.cross_attention_dim)).sample
    assert y.shape == (1, 4, 12, 20)


def test_oracle_pinned_against_diffusers(golden_dir):
    """The UNet / VAE / scheduler restatements against diffusers' own outputs (tests/golden/diffusers_pin.npz, written by
    ``python oracle/pin_against_diffusers.py`` wherever diffusers>=0.25.0 imports - reference requirement,
    /root/reference/requirements.txt:2, call sites marigold_depth_pipeline.py:461-468, 491-492, 512-513).  Until that file
    exists the oracle is PARITY UNPINNED for these modules (structure and parameter counts only): said loudly, not failed."""
    import warnings
    path = os.path.join(golden_dir, "diffusers_pin.npz")
    if not os.path.exists(path):
        msg = ("PARITY UNPINNED: oracle/sd2_unet.py, sd2_vae.py and schedulers.py have not been checked against diffusers "
               "(tests/golden/diffusers_pin.npz is absent; one command where diffusers imports: python oracle/pin_against_diffusers.py)")
        warnings.warn(msg)
        pytest.skip(msg)
    pin = np.load(path)
    assert bool(pin["all_within_tolerance"]), {k: float(pin[k]) for k in pin.files if k.startswith("err/")}
    tol = float(pin["tolerance"])
    usd, vsd = syn.synthetic_unet_state_dict(TINY_UNET), syn.synthetic_vae_state_dict(TINY_VAE)
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    g = torch.Generator().manual_seed(4242)
    with torch.no_grad():
        unet = UNet2DConditionModel(in_channels=TINY_UNET.in_channels, out_channels=TINY_UNET.out_channels,
                                    block_out_channels=TINY_UNET.block_out_channels, attention_head_dim=TINY_UNET.heads,
                                    cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
        unet.load_state_dict(usd)
        x = torch.randn(2, TINY_UNET.in_channels, 8, 16, generator=g)
        for t in (999, 500, 1):
            ref = torch.from_numpy(pin[f"ref/tiny/unet/t{t}"])
            got = unet(x, torch.tensor(t), ctx.expand(2, -1, -1)).sample
            assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), t
        vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels, layers_per_block=TINY_VAE.layers_per_block,
                            latent_channels=TINY_VAE.latent_channels).eval()
        vae.load_state_dict(vsd)
        img = torch.rand(1, 3, 64, 128, generator=g) * 2 - 1
        ref = torch.from_numpy(pin["ref/tiny/vae/encode_moments"])
        got = vae.quant_conv(vae.encoder(img))
        assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())
        ref_dec = torch.from_numpy(pin["ref/tiny/vae/decode"])
        got = vae.decoder(vae.post_quant_conv(ref[:, :TINY_VAE.latent_channels]))
        assert float((got - ref_dec).abs().max()) <= tol * float(ref_dec.abs().max())
