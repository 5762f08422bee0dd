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
    # measured (profiles/r4_*, four builds): rmse 1.65e-3, delta1 0.99998-0.999995 -> bounds at 3x the measured error
    assert m["rmse"] < 5e-3 and m["delta1"] > 0.999 and abs(m["scale"] - 1.0) < 3e-3, m


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


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json's C2 / C4 / C5 and the benchmark batch at 768x768 against the fp32 CPU oracle.  The oracle outputs are
# committed (tests/golden/fullsize_768.npz, minted by oracle/make_fullsize_golden.py: ~25 CPU-minutes that the GPU box
# does not have to spend); the weights are rebuilt here from the same per-tensor seeds.  Tolerances are stated in the
# reference's own metrics: least-squares affine-invariant depth error (src/util/alignment.py:35-82,
# src/util/metric.py:64-104) and mean angular error in degrees for normals (src/util/metric.py:194-223); latents in
# RMSE relative to the RMS of the oracle latent.

# Measured on MI355X (round 2, profiles/r2_parity_fullsize.log): the latent error grows linearly with the step count,
# 4.3e-4 after one DDIM step -> 5.8e-3 after ten (LCM: 8.7e-3 after four, it re-noises every step); depth RMSE
# 1.9e-3 / delta1 0.99999 (C2), 2.2e-3 (C4); normals 0.70 deg per member, 1.37 deg after the closest-member ensemble
# (near-ties between members pick another member: p99 30 deg).  Bounds = ~3x the measured values.
LAT_REL_BOUND = 2e-2       # final x_0 latent: rmse / rms(oracle) after 10 bf16 UNet evaluations (4 for LCM)
DEPTH_RMSE_BOUND, DEPTH_D1_BOUND = 6e-3, 0.999
NORMALS_MEAN_DEG_BOUND = 2.5


@pytest.fixture(scope="module")
def gold768(golden_dir):
    import os
    path = os.path.join(golden_dir, "fullsize_768.npz")
    if not os.path.exists(path):
        pytest.fail("tests/golden/fullsize_768.npz missing: run python -m oracle.make_fullsize_golden")
    return np.load(path)


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), torch.as_tensor(ref).double()
    return float(((got - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


def _stepwise(prog, n_steps, trace, name):
    """Replay a denoising program one step at a time and report the latent error after every step."""
    seq = prog.seq
    per_step = prog.n_fwd_ops       # the scheduler update is the tail of conv_out's pointwise pass
    seq.run_range(0, prog.n_prologue_ops)
    errs = []
    for i in range(n_steps):
        seq.run_range(prog.n_prologue_ops + i * per_step, per_step)
        torch.cuda.synchronize()
        errs.append(_rel(prog.x[:1], trace[i:i + 1]))
    print(f"[parity] {name}: per-step latent rmse/rms vs fp32 oracle: " + " ".join(f"{e:.2e}" for e in errs))
    return errs


def test_c2_depth_768_t10_vs_oracle(full, gold768):
    """C2: depth-v1-1 style (DDIM trailing, zero-SNR, v-prediction), 10 steps, E = 1, 768x768."""
    from marigold_amd import synthetic as syn
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat0 = syn.synthetic_latents(4, 96, 96, seed=7)
    pipe = _pipe(full)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    rl = full["vae"].encode_rgb_latent(rgb.cuda())
    e_enc = _rel(rl, gold768["rgb_latent"])
    print(f"[parity] C2 encode_rgb 768x768: latent rmse/rms {e_enc:.3e}")
    assert e_enc < 2e-2
    pipe.scheduler.set_timesteps(10)
    full["unet"].set_context(full["ctx"])
    prog = full["unet"].denoise_program(1, 96, 96, pipe.scheduler, 10, rgb_broadcast=True)
    prog.rgb_latent.copy_(rl)
    prog.x.copy_(lat0[:1])
    errs = _stepwise(prog, 10, gold768["ddim10_trace_m0"], "C2 DDIM x10 @96x96 latent")
    assert errs[-1] < LAT_REL_BOUND and max(errs) < 2 * LAT_REL_BOUND
    d = pipe.single_infer(rgb, 10, None, False, init_latents=lat0[:1])
    m = omet.affine_invariant_depth_errors(gold768["ddim10_depth_m0"].astype(np.float32), d[0, 0].cpu().numpy())
    print(f"[parity] C2 depth 768x768 T=10 E=1 vs fp32 CPU oracle: {m}")
    assert m["rmse"] < DEPTH_RMSE_BOUND and m["delta1"] > DEPTH_D1_BOUND, m


def test_c4_lcm_768_t4_vs_oracle(full, gold768):
    """C4: LCM, 4 steps, E = 1, 768x768; both sides consume the same per-step noise (CPU generator, seed 99)."""
    import marigold_amd as M
    from marigold_amd import synthetic as syn
    from marigold_amd.schedulers import LCMScheduler
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat0 = syn.synthetic_latents(4, 96, 96, seed=7)
    sched = LCMScheduler()
    pipe = M.MarigoldDepthPipeline(unet=full["unet"], vae=full["vae"], scheduler=sched, empty_text_embed=full["ctx"],
                                   default_denoising_steps=4, default_processing_resolution=768)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    rl = full["vae"].encode_rgb_latent(rgb.cuda())
    full["unet"].set_context(full["ctx"])
    prog = full["unet"].denoise_program(1, 96, 96, sched, 4, rgb_broadcast=True)
    g = torch.Generator("cpu").manual_seed(99)
    assert len(prog.noises) == 3
    for nz in prog.noises:
        nz.copy_(torch.randn((1, 4, 96, 96), generator=g))
    prog.rgb_latent.copy_(rl)
    prog.x.copy_(lat0[:1])
    errs = _stepwise(prog, 4, gold768["lcm4_trace"], "C4 LCM x4 @96x96 latent")
    assert errs[-1] < LAT_REL_BOUND
    d = full["vae"].decode(prog.x, post=1)
    m = omet.affine_invariant_depth_errors(gold768["lcm4_depth"].astype(np.float32), d[0, 0].cpu().numpy())
    print(f"[parity] C4 depth 768x768 LCM T=4 vs fp32 CPU oracle: {m}")
    assert m["rmse"] < DEPTH_RMSE_BOUND and m["delta1"] > DEPTH_D1_BOUND, m
    assert pipe is not None


def test_c5_normals_768_e4_vs_oracle(full, gold768):
    """C5: normals, 10 steps, E = 4, 768x768 through the public call: member latents, member 0 and the ensemble."""
    from marigold_amd import synthetic as syn
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat0 = syn.synthetic_latents(4, 96, 96, seed=7)
    pn = _pipe(full, "normals")
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    n = pn.single_infer(rgb.expand(4, -1, -1, -1), 10, None, False, init_latents=lat0)
    prog = full["unet"].denoise_program(4, 96, 96, pn.scheduler, 10, rgb_broadcast=True)
    for e in range(4):
        r = _rel(prog.x[e:e + 1], gold768["ddim10_final"][e:e + 1])
        print(f"[parity] C5 member {e}: final latent rmse/rms {r:.3e}")
        assert r < LAT_REL_BOUND
    ang = omet.angular_error_deg(n[0].cpu(), gold768["ddim10_normals_m0"].astype(np.float32))
    print(f"[parity] C5 normals member 0: mean {ang.mean():.3f} deg, p99 {np.percentile(ang, 99):.3f} deg")
    assert ang.mean() < NORMALS_MEAN_DEG_BOUND
    out = pn(img, denoising_steps=10, ensemble_size=4, processing_res=0, show_progress_bar=False, init_latents=lat0)
    ang = omet.angular_error_deg(np.asarray(out.normals_np), gold768["ddim10_normals_e4"].astype(np.float32))
    print(f"[parity] C5 ensembled normals E=4: mean {ang.mean():.3f} deg, p99 {np.percentile(ang, 99):.3f} deg")
    assert ang.mean() < NORMALS_MEAN_DEG_BOUND


def test_metric_config_b10_members_vs_oracle(full, gold768):
    """The benchmark's own program (B = 10 members in one batch, 10 steps, 768x768): the four members the oracle
    holds must come out of the 10-batch as they come out of the oracle, and as they come out alone."""
    from marigold_amd import synthetic as syn
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat = torch.cat([syn.synthetic_latents(4, 96, 96, seed=7), syn.synthetic_latents(6, 96, 96, seed=8)])
    pipe = _pipe(full)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    d10 = pipe.single_infer(rgb.expand(10, -1, -1, -1), 10, None, False, init_latents=lat)
    prog = full["unet"].denoise_program(10, 96, 96, pipe.scheduler, 10, rgb_broadcast=True)
    x10 = prog.x.clone()
    for e in range(4):
        r = _rel(x10[e:e + 1], gold768["ddim10_final"][e:e + 1])
        print(f"[parity] metric config member {e} of the B=10 batch: final latent rmse/rms {r:.3e}")
        assert r < LAT_REL_BOUND
    m = omet.affine_invariant_depth_errors(gold768["ddim10_depth_m0"].astype(np.float32), d10[0, 0].cpu().numpy())
    print(f"[parity] metric config member 0 depth vs fp32 CPU oracle: {m}")
    assert m["rmse"] < DEPTH_RMSE_BOUND and m["delta1"] > DEPTH_D1_BOUND, m
    d1 = pipe.single_infer(rgb, 10, None, False, init_latents=lat[:1])
    m1 = omet.affine_invariant_depth_errors(d1[0, 0].cpu().numpy(), d10[0, 0].cpu().numpy())
    print(f"[property] member 0 alone vs inside the B=10 batch (10 steps): {m1}")
    assert m1["rmse"] < 1e-2


def test_c3_batch_e8_members_vs_oracle(full, gold768):
    """C3's per-node total (E = 8 members, 10 steps, 768x768) as ONE batch on one GPU - the program a single-GPU run of C3 executes,
    with its own tile / split-K choices (the tuning table keys on the row count): the four members the oracle holds against the
    oracle, the ensembled map of the eight finite and in range, and member 0 against the B = 10 program's member 0."""
    from marigold_amd import synthetic as syn
    from oracle import metrics as omet
    img = syn.synthetic_image(768, 768, seed=0)
    lat = torch.cat([syn.synthetic_latents(4, 96, 96, seed=7), syn.synthetic_latents(6, 96, 96, seed=8)])
    pipe = _pipe(full)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    d8 = pipe.single_infer(rgb.expand(8, -1, -1, -1), 10, None, False, init_latents=lat[:8])
    prog = full["unet"].denoise_program(8, 96, 96, pipe.scheduler, 10, rgb_broadcast=True)
    for e in range(4):
        r = _rel(prog.x[e:e + 1], gold768["ddim10_final"][e:e + 1])
        print(f"[parity] C3 member {e} of the B=8 batch: final latent rmse/rms {r:.3e}")
        assert r < LAT_REL_BOUND
    m = omet.affine_invariant_depth_errors(gold768["ddim10_depth_m0"].astype(np.float32), d8[0, 0].cpu().numpy())
    print(f"[parity] C3 member 0 depth vs fp32 CPU oracle: {m}")
    assert m["rmse"] < DEPTH_RMSE_BOUND and m["delta1"] > DEPTH_D1_BOUND, m
    out = pipe(img, denoising_steps=10, ensemble_size=8, processing_res=0, show_progress_bar=False, init_latents=lat[:8], color_map=None)
    dn = np.asarray(out.depth_np)
    assert dn.shape == (768, 768) and np.isfinite(dn).all() and dn.min() >= 0.0 and dn.max() <= 1.0
    # the pipeline call is the same eight members through ensemble_depth (reference :294-300): the same map, bit for bit.  (The
    # members of a random-weight model do not agree with one another, so the ensembled map is NOT held against a single member.)
    from marigold_amd.ensemble import ensemble_depth
    ens, _ = ensemble_depth(d8, scale_invariant=True, shift_invariant=True)
    np.testing.assert_allclose(dn, ens.squeeze().cpu().numpy().clip(0, 1), rtol=0, atol=1e-6)


def test_c2_heavy_tailed_weights_vs_oracle(full, golden_dir):
    """C2 on weights with SD-like activation statistics planted (marigold_amd/synthetic.py::plant_heavy_tails: 2 % outlier
    channels at 30x in every layer that writes the transformer residual stream, norm gains at 8x, GEGLU gate rows at 4x) -
    the regime where the folded LayerNorm cancels large numbers, the polynomial GELU is clamped and the fused GroupNorm
    fix-up rounds large values to bf16 - against a freshly minted fp32 oracle golden (oracle/make_fullsize_golden.py
    --heavy), with the LayerNorm fold of the product path: it may lose at most 1.25x against the unfused chain's recorded
    error, else the operand needs centring.  Measured on MI355X (profiles/r3_parity_heavy_tailed_stress.log):
    final latent 2.66e-2 folded vs 2.54e-2 as a pass (1.1e-2 on well-conditioned weights), depth rmse 4.5e-3, delta1 0.99994;
    stated bounds: latent < 4e-2 (2 x LAT_REL_BOUND - outlier channels double the bf16 path's own error), depth as in C2."""
    import os
    from marigold_amd import synthetic as syn
    from marigold_amd.modules import UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler
    from oracle import metrics as omet
    path = os.path.join(golden_dir, "fullsize_768_heavy.npz")
    if not os.path.exists(path):
        pytest.fail("tests/golden/fullsize_768_heavy.npz missing: run python -m oracle.make_fullsize_golden --heavy")
    gold = np.load(path)
    usd = syn.synthetic_unet_state_dict(full["ucfg"], heavy_tail=True)
    img = syn.synthetic_image(768, 768, seed=0)
    lat0 = syn.synthetic_latents(4, 96, 96, seed=7)
    rgb = (img.float() / 255.0 * 2.0 - 1.0)
    rl = full["vae"].encode_rgb_latent(rgb.cuda())
    unet = UNet2DConditionModelHIP(usd, full["ucfg"]).to("cuda:0")
    unet.set_context(full["ctx"])
    sched = DDIMScheduler()
    sched.set_timesteps(10)
    prog = unet.denoise_program(1, 96, 96, sched, 10, rgb_broadcast=True)
    prog.rgb_latent.copy_(rl)
    prog.x.copy_(lat0[:1])
    errs = _stepwise(prog, 10, gold["ddim10_trace_m0"], "C2 heavy-tailed weights, LayerNorm folded")
    final = prog.x.clone()
    del prog, unet
    torch.cuda.empty_cache()
    e_fold = errs[-1]
    # the unfused chain (LayerNorm as its own pass; removed from the engine in round 5) measured 2.54e-2 on the same golden
    # (profiles/r3_parity_heavy_tailed_stress.log): the fold may lose at most 1.25 x against that, else the operand needs
    # centring; the kernel-level form of the comparison stays in tests/test_gpu_kernels.py::test_layernorm_fold_heavy_tailed
    E_PASS_R3 = 2.54e-2
    print(f"[parity] heavy-tailed C2: final latent rmse/rms - folded LayerNorm {e_fold:.3e} (LayerNorm as a pass, round 3: {E_PASS_R3:.3e})")
    assert e_fold < 2 * LAT_REL_BOUND and e_fold <= max(1.25 * E_PASS_R3, LAT_REL_BOUND)
    from marigold_amd import _lib as L
    d = full["vae"].decode(final, post=L.POST_DEPTH)
    m = omet.affine_invariant_depth_errors(gold["ddim10_depth_m0"].astype(np.float32), d[0, 0].float().cpu().numpy())
    print(f"[parity] heavy-tailed C2 depth 768x768 vs fp32 CPU oracle: {m}")
    assert m["rmse"] < DEPTH_RMSE_BOUND and m["delta1"] > DEPTH_D1_BOUND, m
