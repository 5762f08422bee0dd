#!/usr/bin/env python
"""Pin the oracle's UNet / VAE / scheduler restatements against diffusers itself - ONE command, wherever ``import
diffusers`` works (it does not in the build container: diffusers>=0.25.0 is a requirement of the reference,
/root/reference/requirements.txt:2, imported at /root/reference/marigold/marigold_depth_pipeline.py:35-42 and called at
:461-468, 491-492, 512-513, and the image has neither the package nor a network).

    python oracle/pin_against_diffusers.py            # tiny + full SD-v2 / AutoencoderKL configurations
    python oracle/pin_against_diffusers.py --tiny     # tiny configurations only (seconds)

What it does: builds ``diffusers.UNet2DConditionModel`` (SD-v2 config, 8-channel conv_in), ``AutoencoderKL``,
``DDIMScheduler`` and ``LCMScheduler`` with the configurations the Marigold checkpoints ship, loads the SAME seeded
synthetic state dicts the engine's tests use (marigold_amd/synthetic.py - diffusers key scheme, strict=True, so a key or
shape this repo got wrong fails right there), runs both implementations on the same seeded inputs and compares every
stage.  It writes tests/golden/diffusers_pin.npz: the inputs' seeds, diffusers' outputs for the TINY configurations (small
enough to commit) and the measured oracle-vs-diffusers differences of every configuration.  tests/test_oracle_models.py
consumes the file when present (oracle vs the stored diffusers outputs) and reports "parity unpinned" when absent.

TEST INFRASTRUCTURE - nothing under marigold_amd/ imports this.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL = 2e-4   # fp32 vs fp32: summation order only


def unet_kwargs(cfg):
    """diffusers UNet2DConditionModel arguments of the SD-v2 family Marigold fine-tunes (unet/config.json of
    prs-eth/marigold-depth-v1-1; the engine's config_check.py holds the same table)."""
    n = len(cfg.block_out_channels)
    return dict(sample_size=96, in_channels=cfg.in_channels, out_channels=cfg.out_channels, center_input_sample=False,
                flip_sin_to_cos=True, freq_shift=0,
                down_block_types=("CrossAttnDownBlock2D",) * (n - 1) + ("DownBlock2D",),
                mid_block_type="UNetMidBlock2DCrossAttn",
                up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * (n - 1),
                only_cross_attention=False, block_out_channels=tuple(cfg.block_out_channels),
                layers_per_block=cfg.layers_per_block, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu",
                norm_num_groups=cfg.norm_groups, norm_eps=1e-5, cross_attention_dim=cfg.cross_attention_dim,
                attention_head_dim=tuple(cfg.heads), dual_cross_attention=False, use_linear_projection=True,
                upcast_attention=False, resnet_time_scale_shift="default")


def vae_kwargs(cfg):
    n = len(cfg.block_out_channels)
    return dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * n,
                up_block_types=("UpDecoderBlock2D",) * n, block_out_channels=tuple(cfg.block_out_channels),
                layers_per_block=cfg.layers_per_block, act_fn="silu", latent_channels=cfg.latent_channels,
                norm_num_groups=cfg.norm_groups, sample_size=768, scaling_factor=0.18215)


SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
             prediction_type="v_prediction", timestep_spacing="trailing", rescale_betas_zero_snr=True,
             set_alpha_to_one=False, steps_offset=1, clip_sample=False)


def compare(name, got, ref, out, store=False):
    got, ref = got.detach().double(), ref.detach().double()
    scale = float(ref.abs().max().clamp_min(1e-12))
    err = float((got - ref).abs().max()) / scale
    out[f"err/{name}"] = np.float64(err)
    if store:
        out[f"ref/{name}"] = ref.float().numpy()
    print(f"  {name:44s} max|oracle - diffusers| / max|diffusers| = {err:.3e}" + ("" if err <= TOL else "   <-- MISMATCH"))
    return err <= TOL


def pin_models(tag, ucfg, vcfg, out, store, lat_hw, img_hw):
    import diffusers
    from marigold_amd import synthetic as syn
    from oracle.sd2_unet import UNet2DConditionModel as OUNet
    from oracle.sd2_vae import AutoencoderKL as OVAE
    ok = True
    usd, vsd = syn.synthetic_unet_state_dict(ucfg), syn.synthetic_vae_state_dict(vcfg)
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    g = torch.Generator().manual_seed(4242)
    with torch.no_grad():
        d_unet = diffusers.UNet2DConditionModel(**unet_kwargs(ucfg)).eval()
        d_unet.load_state_dict(usd, strict=True)
        o_unet = OUNet(in_channels=ucfg.in_channels, out_channels=ucfg.out_channels, block_out_channels=ucfg.block_out_channels,
                       attention_head_dim=ucfg.heads, cross_attention_dim=ucfg.cross_attention_dim).eval()
        o_unet.load_state_dict(usd)
        x = torch.randn(2, ucfg.in_channels, *lat_hw, generator=g)
        for t in (999, 500, 1):
            ref = d_unet(x, torch.tensor(t), encoder_hidden_states=ctx.expand(2, -1, -1)).sample
            ok &= compare(f"{tag}/unet/t{t}", o_unet(x, torch.tensor(t), ctx.expand(2, -1, -1)).sample, ref, out, store)
        del d_unet, o_unet
        d_vae = diffusers.AutoencoderKL(**vae_kwargs(vcfg)).eval()
        d_vae.load_state_dict(vsd, strict=True)
        o_vae = OVAE(block_out_channels=vcfg.block_out_channels, layers_per_block=vcfg.layers_per_block,
                     latent_channels=vcfg.latent_channels).eval()
        o_vae.load_state_dict(vsd)
        img = torch.rand(1, 3, *img_hw, generator=g) * 2 - 1
        ref_moments = d_vae.quant_conv(d_vae.encoder(img))            # marigold_depth_pipeline.py:491-492
        ok &= compare(f"{tag}/vae/encode_moments", o_vae.quant_conv(o_vae.encoder(img)), ref_moments, out, store)
        z = ref_moments[:, :vcfg.latent_channels]
        ref_dec = d_vae.decoder(d_vae.post_quant_conv(z))              # :512-513
        ok &= compare(f"{tag}/vae/decode", o_vae.decoder(o_vae.post_quant_conv(z)), ref_dec, out, store)
    return ok


def pin_schedulers(out):
    import diffusers
    from oracle.schedulers import DDIMScheduler as ODDIM, LCMScheduler as OLCM
    ok = True
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    for name, T, spacing in (("ddim_trailing_10", 10, "trailing"), ("ddim_trailing_1", 1, "trailing"), ("ddim_leading_50", 50, "leading")):
        cfg = dict(SCHED, timestep_spacing=spacing)
        ds, osch = diffusers.DDIMScheduler(**cfg), ODDIM(**{k: v for k, v in cfg.items() if k not in ("beta_schedule", "clip_sample")})
        ds.set_timesteps(T)
        osch.set_timesteps(T)
        assert list(map(int, ds.timesteps)) == list(map(int, osch.timesteps)), (name, ds.timesteps, osch.timesteps)
        xd, xo = x0.clone(), x0.clone()
        for i, t in enumerate(ds.timesteps):
            mo = torch.randn(x0.shape, generator=torch.Generator().manual_seed(100 + i))
            xd = ds.step(mo, t, xd).prev_sample
            xo = osch.step(mo, t, xo).prev_sample
        ok &= compare(f"sched/{name}", xo, xd, out, True)
    # marigold-depth-lcm-v1-0's scheduler_config.json: no zero-SNR rescale, leading spacing (oracle/schedulers.py defaults)
    lcfg = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="v_prediction",
                timestep_spacing="leading", rescale_betas_zero_snr=False, set_alpha_to_one=False, steps_offset=1)
    ds = diffusers.LCMScheduler(**dict(lcfg, beta_schedule="scaled_linear", clip_sample=False, original_inference_steps=50,
                                       timestep_scaling=10.0))
    osch = OLCM(original_inference_steps=50, timestep_scaling=10.0, **lcfg)
    ds.set_timesteps(4)
    osch.set_timesteps(4)
    assert list(map(int, ds.timesteps)) == list(map(int, osch.timesteps)), (ds.timesteps, osch.timesteps)
    xd, xo = x0.clone(), x0.clone()
    for i, t in enumerate(ds.timesteps):
        mo = torch.randn(x0.shape, generator=torch.Generator().manual_seed(200 + i))
        gd, go = torch.Generator().manual_seed(300 + i), torch.Generator().manual_seed(300 + i)
        xd = ds.step(mo, t, xd, generator=gd).prev_sample
        xo = osch.step(mo, t, xo, generator=go).prev_sample
    ok &= compare("sched/lcm_4", xo, xd, out, True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true", help="tiny configurations only")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "diffusers_pin.npz"))
    args = ap.parse_args()
    try:
        import diffusers
    except ImportError:
        print("parity unpinned: `import diffusers` fails here - run this script where diffusers>=0.25.0 is installed")
        return 2
    from marigold_amd.arch import TINY_UNET, TINY_VAE, UNetConfig, VAEConfig
    torch.set_grad_enabled(False)
    out = {"diffusers_version": np.array(diffusers.__version__), "torch_version": np.array(torch.__version__),
           "tolerance": np.float64(TOL)}
    print(f"diffusers {diffusers.__version__}, torch {torch.__version__}")
    ok = pin_schedulers(out)
    ok &= pin_models("tiny", TINY_UNET, TINY_VAE, out, True, (8, 16), (64, 128))
    if not args.tiny:
        ok &= pin_models("full", UNetConfig(), VAEConfig(), out, False, (24, 24), (192, 192))
    out["all_within_tolerance"] = np.array(bool(ok))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print(("PINNED" if ok else "MISMATCH") + f": wrote {args.out}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
