"""Generate the committed golden fixtures under tests/golden/.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Run in the BUILD container only (it reads
/root/reference, which does not exist on the GPU box):

    python -m oracle.make_golden

1. ``ensemble_ref.npz`` - outputs of the REFERENCE's own marigold/util/ensemble.py
   (imported unmodified from /root/reference with a stub for the two torchvision symbols it
   needs) on seeded inputs.  This pins ``oracle/ensemble.py`` and, through it, the HIP
   ensembling kernels.
2. ``tiny_*.npz`` - outputs of the oracle modules (``oracle/sd2_unet.py``, ``sd2_vae.py``,
   ``schedulers.py``, ``pipeline.py``) on the tiny seeded configuration of
   ``marigold_amd.arch`` - minted here because the reference holds no golden vectors for
   this boundary ("parity unpinned").
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _import_reference_ensemble():
    """Load /root/reference/marigold/util/{image_util,ensemble}.py without diffusers."""
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode:
        BILINEAR, BICUBIC, NEAREST_EXACT = "bilinear", "bicubic", "nearest-exact"

    def resize(img, size, interpolation="bilinear", antialias=True):
        aa = antialias and interpolation in ("bilinear", "bicubic")
        return torch.nn.functional.interpolate(img, size=tuple(size), mode=interpolation,
                                               antialias=aa, **({} if "nearest" in interpolation
                                                                else {"align_corners": False}))

    tvt.InterpolationMode = InterpolationMode
    tvf.resize = resize
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})
    pkg = types.ModuleType("refmarigold_util")
    pkg.__path__ = [os.path.join(REF, "marigold", "util")]
    sys.modules["refmarigold_util"] = pkg
    mods = {}
    for name in ("image_util", "ensemble"):
        spec = importlib.util.spec_from_file_location(
            f"refmarigold_util.{name}", os.path.join(REF, "marigold", "util", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["ensemble"]


def synth_depth_members(E, H, W, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(1, 1, 6, 8, generator=g), size=(H, W),
                                           mode="bicubic", align_corners=False).clamp(0, 1)
    a = 0.6 + 0.5 * torch.rand(E, 1, 1, 1, generator=g)
    b = 0.2 * torch.rand(E, 1, 1, 1, generator=g)
    noise = 0.03 * torch.randn(E, 1, H, W, generator=g)
    return (a * base + b + noise).clamp(0, 1).contiguous()


def synth_realistic_depth_members(E, H, W, seed):
    """Members shaped like real Marigold predictions: every member is a [0,1]-normalised view of the
    same scene (single_infer ends with clip + (x+1)/2, marigold_depth_pipeline.py:473-475) with a
    slightly different affine gauge, low-frequency disagreement and ~1 % pixel noise."""
    g = torch.Generator("cpu").manual_seed(seed)
    up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bicubic", align_corners=False)
    base = up(torch.rand(1, 1, 6, 8, generator=g))
    base = (base - base.min()) / (base.max() - base.min())
    a = 0.90 + 0.10 * torch.rand(E, 1, 1, 1, generator=g)
    b = 0.04 * torch.rand(E, 1, 1, 1, generator=g)
    wobble = 0.03 * up(torch.randn(E, 1, 4, 5, generator=g))
    noise = 0.01 * torch.randn(E, 1, H, W, generator=g)
    return (a * base + b + wobble + noise).clamp(0, 1).contiguous()


def synth_normal_members(E, H, W, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.randn(1, 3, 5, 7, generator=g), size=(H, W),
                                           mode="bicubic", align_corners=False)
    n = base + 0.25 * torch.randn(E, 3, H, W, generator=g)
    return (n / n.norm(dim=1, keepdim=True).clamp(min=1e-6)).contiguous()


def make_ensemble_golden():
    ref = _import_reference_ensemble()
    out = {}
    cases = [("d_e4", 4, 48, 64, 11), ("d_e10", 10, 32, 40, 12), ("d_e3", 3, 24, 24, 13)]
    for name, E, H, W, seed in cases:
        x = synth_depth_members(E, H, W, seed)
        d, u = ref.ensemble_depth(x.clone(), True, True, output_uncertainty=True)
        out[f"{name}_in"], out[f"{name}_out"], out[f"{name}_unc"] = x.numpy(), d.numpy(), u.numpy()
    for name, E, H, W, seed in [("d_real_e10", 10, 96, 128, 31), ("d_real_e4", 4, 64, 64, 32)]:
        x = synth_realistic_depth_members(E, H, W, seed)
        d, u = ref.ensemble_depth(x.clone(), True, True, output_uncertainty=True)
        out[f"{name}_in"], out[f"{name}_out"], out[f"{name}_unc"] = x.numpy(), d.numpy(), u.numpy()
    x = synth_depth_members(4, 24, 32, 14)
    d, u = ref.ensemble_depth(x.clone(), True, False, output_uncertainty=True, reduction="mean")
    out["d_scale_mean_in"], out["d_scale_mean_out"], out["d_scale_mean_unc"] = \
        x.numpy(), d.numpy(), u.numpy()
    # NB: (scale_invariant=False, shift_invariant=False) raises ValueError in the reference
    # (ensemble.py:189-190) despite the docstring; tests pin that as error behaviour.
    try:
        ref.ensemble_depth(synth_depth_members(5, 16, 16, 15), False, False)
        out["abs_raises"] = np.array(0)
    except ValueError:
        out["abs_raises"] = np.array(1)
    for name, E, H, W, seed in [("n_e4", 4, 40, 56, 21), ("n_e10", 10, 24, 24, 22)]:
        x = synth_normal_members(E, H, W, seed)
        n, u = ref.ensemble_normals(x.clone(), output_uncertainty=True)
        m, _ = ref.ensemble_normals(x.clone(), reduction="mean")
        out[f"{name}_in"], out[f"{name}_closest"], out[f"{name}_unc"], out[f"{name}_mean"] = \
            x.numpy(), n.numpy(), u.numpy(), m.numpy()
    g = torch.Generator("cpu").manual_seed(41)
    x = torch.rand(5, 6, 20, 24, generator=g)       # E=5 members, 2 modalities x 3 channels
    for red in ("median", "mean"):
        p, u = ref.ensemble_iid(x.clone(), output_uncertainty=True, reduction=red)
        out[f"iid_{red}_pred"], out[f"iid_{red}_unc"] = p.numpy(), u.numpy()
    out["iid_in"] = x.numpy()
    np.savez_compressed(os.path.join(GOLD, "ensemble_ref.npz"), **out)
    print("ensemble_ref.npz:", sorted(out))


def make_ensemble_768_golden():
    """The metric configuration of test-time ensembling: E = 10 members at 768 x 768 through the REFERENCE's own
    ensemble_depth (about two minutes of CPU: ~4 600 cost evaluations).  The members come from the seeded
    generator above (the test regenerates them), only the reference's outputs are stored."""
    import time
    ref = _import_reference_ensemble()
    x = synth_realistic_depth_members(10, 768, 768, 51)
    t0 = time.time()
    d, u = ref.ensemble_depth(x.clone(), True, True, output_uncertainty=True)
    dt = time.time() - t0
    np.savez_compressed(os.path.join(GOLD, "ensemble_ref_768.npz"), d_real_e10_768_out=d.numpy()[0, 0],
                        d_real_e10_768_unc=u.numpy()[0, 0].astype(np.float16), seconds_reference_cpu=np.float64(dt),
                        threads=np.int64(torch.get_num_threads()))
    print(f"ensemble_ref_768.npz: reference ensemble_depth E=10 768x768 took {dt:.1f} s on {torch.get_num_threads()} threads")


def make_tiny_golden():
    sys.path.insert(0, ROOT)
    from marigold_amd.arch import TINY_UNET, TINY_VAE
    from marigold_amd import synthetic as syn
    from oracle import pipeline as opipe
    from oracle.schedulers import DDIMScheduler, LCMScheduler
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL

    torch.manual_seed(0)
    torch.set_num_threads(8)
    unet = UNet2DConditionModel(block_out_channels=TINY_UNET.block_out_channels,
                                attention_head_dim=TINY_UNET.heads,
                                cross_attention_dim=TINY_UNET.cross_attention_dim).eval()
    unet.load_state_dict(syn.synthetic_unet_state_dict(TINY_UNET))
    vae = AutoencoderKL(block_out_channels=TINY_VAE.block_out_channels).eval()
    vae.load_state_dict(syn.synthetic_vae_state_dict(TINY_VAE))
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    img = syn.synthetic_image(64, 128, seed=0)          # -> latent 8 x 16
    lat0 = syn.synthetic_latents(3, 8, 16, seed=2024)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    out = {}
    with torch.no_grad():
        rgb_lat = opipe.encode_rgb(vae, rgb)
        out["rgb_latent"] = rgb_lat.numpy()
        x8 = torch.cat([rgb_lat.expand(3, -1, -1, -1), lat0], dim=1)
        out["unet_t999"] = unet(x8, torch.tensor(999), ctx.repeat(3, 1, 1)).sample.numpy()
        out["unet_t249"] = unet(x8, torch.tensor(249), ctx.repeat(3, 1, 1)).sample.numpy()
        out["decoded"] = opipe.decode_latent(vae, lat0 * 0.5).numpy()
        # yardstick for the bf16 engine: the SAME oracle modules run in bf16 on the CPU (what the
        # reference's reduced-precision path would do); only the rmse vs fp32 is stored.
        import copy
        u16 = copy.deepcopy(unet).to(torch.bfloat16)
        v16 = copy.deepcopy(vae).to(torch.bfloat16)
        bf = torch.bfloat16

        def rm(a, b):
            return np.float64(((a.double() - torch.from_numpy(b).double()) ** 2).mean().sqrt())
        out["bf16_rmse_rgb_latent"] = rm(opipe.encode_rgb(v16, rgb.to(bf)), out["rgb_latent"])
        for t in (999, 249):
            y16 = u16(x8.to(bf), torch.tensor(t), ctx.repeat(3, 1, 1).to(bf)).sample
            out[f"bf16_rmse_unet_t{t}"] = rm(y16, out[f"unet_t{t}"])
        out["bf16_rmse_decoded"] = rm(opipe.decode_latent(v16, (lat0 * 0.5).to(bf)), out["decoded"])
        # depth v1-1 style: DDIM trailing + zero SNR, v-pred, 4 steps, E=3
        sch = DDIMScheduler()
        tr = []
        d = opipe.single_infer("depth", unet, vae, sch, rgb.expand(3, -1, -1, -1), lat0, ctx, 4,
                               trace=tr)
        out["depth_ddim4"] = d.numpy()
        out["depth_ddim4_latents"] = np.stack([x[2].numpy() for x in tr])
        n = opipe.single_infer("normals", unet, vae, DDIMScheduler(), rgb.expand(3, -1, -1, -1),
                               lat0, ctx, 2)
        out["normals_ddim2"] = n.numpy()
        # v1-0 style: leading, no zero-SNR, epsilon... (checkpoint uses v-pred; keep v)
        d10 = opipe.single_infer("depth", unet, vae,
                                 DDIMScheduler(timestep_spacing="leading",
                                               rescale_betas_zero_snr=False),
                                 rgb, lat0[:1], ctx, 3)
        out["depth_leading3"] = d10.numpy()
        # LCM, 3 steps, per-step noise from a CPU generator seeded 99
        g = torch.Generator("cpu").manual_seed(99)
        dl = opipe.single_infer("depth", unet, vae, LCMScheduler(), rgb, lat0[:1], ctx, 3,
                                generator=g)
        out["depth_lcm3"] = dl.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_pipeline.npz"), **out)
    print("tiny_pipeline.npz:", {k: v.shape for k, v in out.items()})


def make_scheduler_golden():
    """Closed-form tables the reference relies on."""
    out = {
        "ddim_trailing_10": np.array([999, 899, 799, 699, 599, 499, 399, 299, 199, 99]),
        "ddim_trailing_4": np.array([999, 749, 499, 249]),
        "ddim_trailing_1": np.array([999]),
        "ddim_leading_10": np.array([901, 801, 701, 601, 501, 401, 301, 201, 101, 1]),
        "ddim_leading_1": np.array([1]),
        "lcm_4": np.array([999, 759, 499, 259]),
        "lcm_1": np.array([999]),
    }
    np.savez(os.path.join(GOLD, "scheduler_tables.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "ensemble768":
        make_ensemble_768_golden()
        sys.exit(0)
    make_scheduler_golden()
    make_ensemble_golden()
    make_tiny_golden()
