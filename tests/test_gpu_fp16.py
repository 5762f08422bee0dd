"""The fp16-operand build of the engine (libmarigold_hip_f16.so: the same kernel sources compiled with OPERAND_F16=1) - what
``from_pretrained(torch_dtype=torch.float16)`` / ``--fp16`` selects, i.e. the arithmetic of the reference's own half-precision path
(script/depth/run.py:203-211; marigold_depth_pipeline.py:253, 433 run the whole model in ``self.dtype``): fp16 operands, fp32
accumulation.  Checked at three levels: every kernel test of tests/test_gpu_kernels.py re-run on fp16 operands; the tiny pipeline
against the committed goldens (bounds: those of the bf16 build - fp16 has three more mantissa bits and must not be worse); BASELINE's
C2 (768x768, 10 DDIM steps) on the full SD-v2 architecture against the fp32 CPU oracle, the latent error printed beside bf16's.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_suite_on_fp16_operands():
    """tests/test_gpu_kernels.py with MARIGOLD_TEST_OPERANDS=fp16: operands rounded to fp16, launches through the fp16 library,
    fp32 torch references of the rounded values, the same tolerances.  (Cases built on bf16's exponent range are marked
    ``bf16_only`` there and skip.)"""
    env = dict(os.environ, MARIGOLD_TEST_OPERANDS="fp16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    print(tail)
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert " passed" in tail


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), torch.as_tensor(ref).double()
    return float(((got - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


def test_tiny_pipeline_fp16_vs_goldens(golden_dir):
    """Stage by stage on the tiny architecture: VAE encode, UNet forward, VAE decode and the public depth call (E = 3, ensembling)
    of the fp16 build against the oracle goldens, each next to the bf16 build's error on the same input."""
    import marigold_amd as M
    from marigold_amd import _lib as L, synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler
    from oracle import metrics as omet
    gold = np.load(os.path.join(golden_dir, "tiny_pipeline.npz"))
    usd, vsd = syn.synthetic_unet_state_dict(TINY_UNET), syn.synthetic_vae_state_dict(TINY_VAE)
    ctx = syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim)
    img = syn.synthetic_image(64, 128, seed=0)
    lat0 = syn.synthetic_latents(3, 8, 16, seed=2024)
    rgb = img.float() / 255.0 * 2.0 - 1.0
    rl = torch.from_numpy(gold["rgb_latent"])
    x8 = torch.cat([rl.expand(3, -1, -1, -1), lat0], dim=1)
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        unet = UNet2DConditionModelHIP(usd, TINY_UNET, compute_dtype=dt).to("cuda:0")
        vae = AutoencoderKLHIP(vsd, TINY_VAE, compute_dtype=dt).to("cuda:0")
        assert unet.dtype == dt and vae.f16 == (dt == torch.float16)
        enc = vae.encode_rgb_latent(rgb.cuda())
        dec = vae.decode((lat0 * 0.5).cuda(), post=L.POST_NONE)
        eps = unet(x8.cuda(), 249, ctx.repeat(3, 1, 1)).sample
        res[name] = dict(enc=_rel(enc, gold["rgb_latent"]), dec=_rel(dec, gold["decoded"]), unet=_rel(eps, gold["unet_t249"]))
        pipe = M.MarigoldDepthPipeline(unet=unet, vae=vae, scheduler=DDIMScheduler(), empty_text_embed=ctx, default_denoising_steps=2,
                                       default_processing_resolution=0)
        assert pipe.dtype == dt
        res[name]["single"] = pipe.single_infer(rgb[None] if rgb.dim() == 3 else rgb, 2, None, False, init_latents=lat0[:1])[0, 0].cpu().numpy()
        g = torch.Generator(device="cuda:0").manual_seed(11)
        res[name]["depth"] = pipe(img, denoising_steps=2, ensemble_size=3, processing_res=0, color_map=None, show_progress_bar=False,
                                  generator=g).depth_np
    print("[parity] tiny pipeline, rmse / rms(oracle):  " +
          "  ".join(f"{k}: bf16 {res['bf16'][k]:.2e} fp16 {res['fp16'][k]:.2e}" for k in ("enc", "unet", "dec")))
    for k in ("enc", "unet", "dec"):
        assert res["fp16"][k] <= max(1.25 * res["bf16"][k], 2e-3), (k, res["fp16"][k], res["bf16"][k])
    for name in ("bf16", "fp16"):   # the public call (three members, ensembling): a map in range from both builds
        d = res[name]["depth"]
        assert d.shape == (64, 128) and np.isfinite(d).all() and d.min() >= 0.0 and d.max() <= 1.0
    m = omet.affine_invariant_depth_errors(res["bf16"]["single"], res["fp16"]["single"])
    print(f"[parity] tiny single_infer (2 steps, given latents): fp16 build vs bf16 build: {m}")
    assert m["rmse"] < 3e-2


def test_c2_fp16_768_t10_vs_oracle(golden_dir):
    """BASELINE C2 (768x768, 10 DDIM steps, E = 1) on the full SD-v2 architecture with fp16 operands against the fp32 CPU oracle
    (tests/golden/fullsize_768.npz): latent after every step and the depth map in the reference's metrics - bf16 measures 6.1e-3 /
    RMSE 1.9e-3 (tests/test_gpu_fullsize.py); fp16 must not be worse than bf16's bound and is expected well inside it.  Then the
    benchmark's batch of ten members (the decoder's 512-wide flash attention, the key-split flash launches, the 4-wave tiles)."""
    import marigold_amd as M
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import UNetConfig, VAEConfig
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler
    from marigold_amd.util.host import usable_cores
    from oracle import metrics as omet
    path = os.path.join(golden_dir, "fullsize_768.npz")
    if not os.path.exists(path):
        pytest.fail("tests/golden/fullsize_768.npz missing: run python -m oracle.make_fullsize_golden")
    gold = np.load(path)
    torch.set_num_threads(min(32, usable_cores()))
    ucfg, vcfg = UNetConfig(), VAEConfig()
    unet = UNet2DConditionModelHIP(syn.synthetic_unet_state_dict(ucfg), ucfg, compute_dtype=torch.float16).to("cuda:0")
    vae = AutoencoderKLHIP(syn.synthetic_vae_state_dict(vcfg), vcfg, compute_dtype=torch.float16).to("cuda:0")
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    pipe = M.MarigoldDepthPipeline(unet=unet, vae=vae, scheduler=DDIMScheduler(), empty_text_embed=ctx, default_denoising_steps=10,
                                   default_processing_resolution=768)
    img = syn.synthetic_image(768, 768, seed=0)
    lat = torch.cat([syn.synthetic_latents(4, 96, 96, seed=7), syn.synthetic_latents(6, 96, 96, seed=8)])
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    rl = vae.encode_rgb_latent(rgb.cuda())
    e_enc = _rel(rl, gold["rgb_latent"])
    print(f"[parity] fp16 C2 encode_rgb 768x768: latent rmse/rms {e_enc:.3e}")
    assert e_enc < 2e-2
    pipe.scheduler.set_timesteps(10)
    unet.set_context(ctx)
    prog = unet.denoise_program(1, 96, 96, pipe.scheduler, 10, rgb_broadcast=True)
    prog.rgb_latent.copy_(rl)
    prog.x.copy_(lat[:1])
    seq = prog.seq
    seq.run_range(0, prog.n_prologue_ops)
    errs = []
    for i in range(10):
        seq.run_range(prog.n_prologue_ops + i * prog.n_fwd_ops, prog.n_fwd_ops)
        torch.cuda.synchronize()
        errs.append(_rel(prog.x[:1], gold["ddim10_trace_m0"][i:i + 1]))
    print("[parity] fp16 C2 DDIM x10 @96x96 latent: per-step rmse/rms vs fp32 oracle: " + " ".join(f"{e:.2e}" for e in errs) +
          "   (bf16 build: 4.2e-04 ... 6.1e-03)")
    assert errs[-1] < 2e-2 and max(errs) < 4e-2
    d = pipe.single_infer(rgb, 10, None, False, init_latents=lat[:1])
    m = omet.affine_invariant_depth_errors(gold["ddim10_depth_m0"].astype(np.float32), d[0, 0].cpu().numpy())
    print(f"[parity] fp16 C2 depth 768x768 T=10 E=1 vs fp32 CPU oracle: {m}   (bf16 build: rmse 1.9e-3)")
    assert m["rmse"] < 6e-3 and m["delta1"] > 0.999, m
    # the benchmark batch: ten members in one program, members 0-3 against the oracle
    d10 = pipe.single_infer(rgb.expand(10, -1, -1, -1), 10, None, False, init_latents=lat)
    p10 = unet.denoise_program(10, 96, 96, pipe.scheduler, 10, rgb_broadcast=True)
    for e in range(4):
        r = _rel(p10.x[e:e + 1], gold["ddim10_final"][e:e + 1])
        print(f"[parity] fp16 metric config member {e} of the B=10 batch: final latent rmse/rms {r:.3e}")
        assert r < 2e-2
    m = omet.affine_invariant_depth_errors(gold["ddim10_depth_m0"].astype(np.float32), d10[0, 0].cpu().numpy())
    print(f"[parity] fp16 metric config member 0 depth vs fp32 CPU oracle: {m}")
    assert m["rmse"] < 6e-3 and m["delta1"] > 0.999, m
    out = pipe(img, denoising_steps=10, ensemble_size=10, processing_res=0, show_progress_bar=False, init_latents=lat, color_map=None)
    dn = np.asarray(out.depth_np)
    assert dn.shape == (768, 768) and np.isfinite(dn).all() and dn.min() >= 0.0 and dn.max() <= 1.0
