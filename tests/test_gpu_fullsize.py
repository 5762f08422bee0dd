"""Full-architecture checks on a real MI355X (SD-v2 UNet 865.9 M + AutoencoderKL, seeded synthetic
weights): BASELINE.json's C1 configuration against the CPU oracle, and size-independent properties at
the benchmark resolution (768x768) where the oracle would take minutes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    assert torch.cuda.is_available()
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import UNetConfig, VAEConfig
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(min(32, usable_cores()))
    ucfg, vcfg = UNetConfig(), VAEConfig()
    usd, vsd = syn.synthetic_unet_state_dict(ucfg), syn.synthetic_vae_state_dict(vcfg)
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    unet = UNet2DConditionModelHIP(usd, ucfg).to("cuda:0")
    vae = AutoencoderKLHIP(vsd, vcfg).to("cuda:0")
    return dict(usd=usd, vsd=vsd, ctx=ctx, unet=unet, vae=vae, ucfg=ucfg, vcfg=vcfg)


def _pipe(full, kind="depth", **sched_kw):
    import marigold_amd as M
    from marigold_amd.schedulers import DDIMScheduler
    cls = M.MarigoldDepthPipeline if kind == "depth" else M.MarigoldNormalsPipeline
    return cls(unet=full["unet"], vae=full["vae"], scheduler=DDIMScheduler(**sched_kw), empty_text_embed=full["ctx"],
               default_denoising_steps=1, default_processing_resolution=768)


def test_c1_config_vs_cpu_oracle(full):
    """C1: depth-v1-0 style (DDIM leading, no zero-SNR), 1 step, E=1, 384x512 at native resolution:
    full-size engine vs the fp32 CPU oracle in the reference's affine-invariant metrics."""
    from marigold_amd import synthetic as syn
    from oracle import metrics as omet, pipeline as opipe
    from oracle.schedulers import DDIMScheduler as ODDIM
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL
    img = syn.synthetic_image(384, 512, seed=0)
    lat0 = syn.synthetic_latents(1, 48, 64, seed=2024)
    kw = dict(timestep_spacing="leading", rescale_betas_zero_snr=False)
    pipe = _pipe(full, "depth", **kw)
    out = pipe(img, denoising_steps=1, ensemble_size=1, processing_res=0, color_map=None, show_progress_bar=False,
               init_latents=lat0)
    assert out.depth_np.shape == (384, 512) and np.isfinite(out.depth_np).all()
    ounet = UNet2DConditionModel().eval()
    ounet.load_state_dict(full["usd"])
    ovae = AutoencoderKL().eval()
    ovae.load_state_dict(full["vsd"])
    ref, _, _ = opipe.predict("depth", ounet, ovae, ODDIM(**kw), img, lat0, full["ctx"], 1)
    m = omet.affine_invariant_depth_errors(ref.squeeze().numpy(), out.depth_np)
    print(f"[parity] C1 full-size depth 384x512 T=1 E=1 vs fp32 CPU oracle: {m}")
    assert m["rmse"] < 0.03 and m["delta1"] > 0.95, m


def test_768_properties(full):
    """Benchmark resolution: determinism, batch invariance, range / unit-norm invariants, identical
    members ensemble to themselves, C1-style resize-in / resize-out shapes."""
    from marigold_amd import ensemble as ens, synthetic as syn
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat0 = syn.synthetic_latents(3, 96, 96, seed=7)
    pipe = _pipe(full)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    a = pipe.single_infer(rgb.expand(3, -1, -1, -1), 2, None, False, init_latents=lat0)
    b = pipe.single_infer(rgb.expand(3, -1, -1, -1), 2, None, False, init_latents=lat0)
    assert a.shape == (3, 1, 768, 768) and torch.isfinite(a).all()
    assert torch.equal(a, b), "same inputs must give bit-identical predictions"
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    solo = pipe.single_infer(rgb, 2, None, False, init_latents=lat0[1:2])
    m = omet.affine_invariant_depth_errors(solo[0, 0].cpu().numpy(), a[1, 0].cpu().numpy())
    print(f"[property] member 1 alone vs inside a batch of 3: {m}")
    assert m["rmse"] < 5e-3, "members are independent: batching must not change a prediction"
    same = a[:1].expand(4, -1, -1, -1).contiguous()
    d, u = ens.ensemble_depth(same, True, True, output_uncertainty=True)
    want = (a[:1] - a[:1].min()) / (a[:1].max() - a[:1].min())
    assert float((d - want).abs().max()) < 1e-5 and float(u.abs().max()) < 1e-6
    pn = _pipe(full, "normals")
    n = pn.single_infer(rgb, 1, None, False, init_latents=lat0[:1])
    assert n.shape == (1, 3, 768, 768) and float((n.norm(dim=1) - 1).abs().max()) < 1e-3
    small = syn.synthetic_image(384, 512, seed=1)
    out = pipe(small, denoising_steps=1, ensemble_size=2, processing_res=768, color_map=None,
               show_progress_bar=False)   # C1 input up-scaled to 576x768 -> latent 72x96, resized back
    assert out.depth_np.shape == (384, 512) and out.depth_np.min() >= 0 and out.depth_np.max() <= 1
