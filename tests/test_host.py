"""CPU-only checks of the host logic: C-ABI library loads and exports what include/*.h declares,
scheduler coefficients vs the oracle schedulers, weight re-layout algebra, resize/batch-size
policy, pipeline error behaviour, member-parallel gather over gloo (world_size 2)."""
import logging
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    from marigold_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "marigold_hip.h")).read()
    body = hdr[hdr.index("typedef struct mg_program"):]
    names = sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", body)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in marigold_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert lib.mg_abi_version() == _lib.ABI_VERSION
    import ctypes
    assert ctypes.sizeof(_lib.MgOp) == 360   # kind + i[40] + f[8] (+4 pad) + p[16] + l[4]


def test_missing_library_fails_loudly(monkeypatch):
    from marigold_amd import _lib
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmarigold_hip.so")
    monkeypatch.setattr(_lib, "LIB_PATH_F16", "/nonexistent/libmarigold_hip_f16.so")
    with pytest.raises(_lib.MarigoldHipError):
        _lib.load()
    with pytest.raises(_lib.MarigoldHipError):
        _lib.load(f16=True)


def test_fp16_engine_programs_validate_without_gpu():
    """compute_dtype=torch.float16: the modules pack fp16 weights, their programs belong to the fp16 library and pass ITS contract
    checks (tiny UNet + VAE, host buffers); bf16 stays the default; other dtypes are refused."""
    from marigold_amd import weights as Wm
    from marigold_amd.arch import TINY_UNET, TINY_VAE, unet_param_shapes, vae_param_shapes
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler
    usd = {k: torch.zeros(s) for k, s in unet_param_shapes(TINY_UNET).items()}
    vsd = {k: torch.zeros(s) for k, s in vae_param_shapes(TINY_VAE).items()}
    unet = UNet2DConditionModelHIP(usd, TINY_UNET, compute_dtype=torch.float16).dry()
    unet.set_context(torch.zeros(1, 2, TINY_UNET.cross_attention_dim))
    prog = unet.denoise_program(2, 8, 16, DDIMScheduler(), 2)
    assert prog.seq.f16 and unet.dtype == torch.float16 and unet.ws.dtype == torch.float16
    prog.seq.validate()
    assert any(t.dtype == torch.float16 for v in unet.ws.cache.values() for t in (v if isinstance(v, tuple) else (v,)) if torch.is_tensor(t))
    assert not any(t.dtype == torch.bfloat16 for v in unet.ws.cache.values() for t in (v if isinstance(v, tuple) else (v,)) if torch.is_tensor(t))
    vae = AutoencoderKLHIP(vsd, TINY_VAE, compute_dtype=torch.float16).dry()
    seq, _, _ = vae._program("decode", 2, 8, 16, 1)
    assert seq.f16
    seq.validate()
    assert not UNet2DConditionModelHIP(usd, TINY_UNET).dry().denoise_program.__self__.f16
    with pytest.raises(ValueError):
        UNet2DConditionModelHIP(usd, TINY_UNET, compute_dtype=torch.float32)
    big = torch.tensor([1e6, -1e6, 1.0])
    assert Wm.to_op16(big, torch.float16).tolist() == [65504.0, -65504.0, 1.0] and Wm.to_op16(big).dtype == torch.bfloat16


def test_both_operand_builds_export_the_abi():
    """libmarigold_hip.so (bf16 operands) and libmarigold_hip_f16.so (fp16: the reference's --fp16 arithmetic) come from the same
    sources and export every symbol of include/marigold_hip.h; each says which operand type it was built for, and the loader refuses
    a swapped pair."""
    from marigold_amd import _lib
    a, b = _lib.load(), _lib.load(f16=True)
    assert a is not b and a.mg_operand_bits() == 0 and b.mg_operand_bits() == 1
    assert a.mg_abi_version() == b.mg_abi_version() == _lib.ABI_VERSION
    for name in _lib.EXPORTS:
        assert hasattr(a, name) and hasattr(b, name), name
    saved = dict(_lib._libs)
    try:
        _lib._libs.clear()
        old = _lib.LIB_PATH, _lib.LIB_PATH_F16
        _lib.LIB_PATH, _lib.LIB_PATH_F16 = old[1], old[0]
        with pytest.raises(_lib.MarigoldHipError):
            _lib.load()
    finally:
        _lib.LIB_PATH, _lib.LIB_PATH_F16 = old
        _lib._libs.clear()
        _lib._libs.update(saved)


@pytest.mark.parametrize("kind,kw,n", [
    ("ddim", dict(), 10), ("ddim", dict(), 4), ("ddim", dict(), 1),
    ("ddim", dict(timestep_spacing="leading", rescale_betas_zero_snr=False), 10),
    ("ddim", dict(timestep_spacing="leading", rescale_betas_zero_snr=False, prediction_type="epsilon"), 5),
    ("ddim", dict(prediction_type="sample"), 3),
    ("lcm", dict(), 4), ("lcm", dict(), 1), ("lcm", dict(prediction_type="epsilon"), 3)])
def test_scheduler_coefficients_match_oracle_step(kind, kw, n):
    from marigold_amd import schedulers as S
    from oracle import schedulers as OS
    mine = (S.DDIMScheduler if kind == "ddim" else S.LCMScheduler)(**kw)
    orc = (OS.DDIMScheduler if kind == "ddim" else OS.LCMScheduler)(**kw)
    mine.set_timesteps(n)
    orc.set_timesteps(n)
    assert mine.timesteps.tolist() == orc.timesteps.tolist()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    for i, t in enumerate(orc.timesteps):
        v = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
        gen = torch.Generator().manual_seed(100 + i)
        ref = orc.step(v, t, x, generator=gen).prev_sample
        cx, cm, cn = mine.step_coefficients(i)
        got = cx * x + cm * v
        if mine.needs_noise(i):
            gen2 = torch.Generator().manual_seed(100 + i)
            got = got + cn * torch.randn(v.shape, generator=gen2, dtype=torch.float64)
        torch.testing.assert_close(got, ref, atol=2e-5, rtol=1e-5)
        x = ref


@pytest.mark.parametrize("group", [32])
def test_geglu_interleave_is_a_permutation_with_paired_rows(group):
    from marigold_amd import _lib as L, weights as Wm
    assert L.load().mg_geglu_interleave() == 32
    C = 16
    w = torch.arange(8 * C * 3, dtype=torch.float32).reshape(8 * C, 3)
    b = torch.arange(8 * C, dtype=torch.float32)
    wp, bp = Wm.pack_geglu(w, b, group)
    assert sorted(bp.tolist()) == b.tolist()
    h = group // 2
    for blk in range(8 * C // group):
        for j in range(h):
            assert bp[group * blk + j] == h * blk + j                # u row
            assert bp[group * blk + h + j] == 4 * C + h * blk + j    # its gate row
    assert torch.equal(wp[:, 0] / 3, bp)


def test_cross_attention_collapse_equals_attention():
    from marigold_amd import weights as Wm
    from oracle.sd2_unet import Attention
    torch.manual_seed(0)
    C, heads, cross = 128, 2, 64
    att = Attention(C, heads, C // heads, cross_dim=cross).double()
    ctx = torch.randn(2, cross, dtype=torch.float64)
    y = torch.randn(50, C, dtype=torch.float64)
    ref = att(y[None], ctx[None])[0]
    wqk, vot, npad = Wm.cross_attention_tables(att.to_q.weight, att.to_k.weight, att.to_v.weight,
                                               att.to_out[0].weight, ctx, heads)
    s = (y @ wqk.double().t())[:, :2 * heads] * (C // heads) ** -0.5
    p = torch.softmax(s.reshape(50, heads, 2), dim=-1).reshape(50, 2 * heads)
    out = p @ vot.double()[:, :2 * heads].t() + att.to_out[0].bias
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-5)
    assert npad == 64 and (wqk[2 * heads:] == 0).all()


def test_conv_pack_layout():
    from marigold_amd import weights as Wm
    w = torch.randn(5, 64, 3, 3)
    x = torch.randn(1, 64, 6, 6)
    cols = F.unfold(x, 3, padding=1)                         # [1, 64*9, 36], index c*9 + tap
    cols = cols.reshape(1, 64, 9, 36).permute(0, 2, 1, 3).reshape(1, 9 * 64, 36)   # tap*64 + c
    out = Wm.pack_conv3x3(w) @ cols[0]
    torch.testing.assert_close(out.reshape(5, 6, 6), F.conv2d(x, w, padding=1)[0], atol=1e-4, rtol=1e-4)


def test_subpixel_upsample_weights_equal_upsample_then_conv():
    """nearest-2x + conv3x3 == four 2x2 convolutions on the low-resolution input with pre-summed taps."""
    from marigold_amd import weights as Wm
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 5, 3, 3, generator=g, dtype=torch.float64)
    x = torch.randn(2, 5, 7, 4, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    ws = Wm.pack_conv3x3_subpix(w).reshape(4, 6, 2, 2, 5)          # [z][n][ty][tx][c]
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            k = ws[2 * a + b].permute(0, 3, 1, 2)                      # [n][c][ty][tx]
            xp = F.pad(x, (1 - b, b, 1 - a, a))                        # window rows y-1+a.., cols x-1+b..
            out[:, :, a::2, b::2] = F.conv2d(xp, k)
    torch.testing.assert_close(out, ref, atol=1e-12, rtol=1e-12)


def test_image_util_and_batchsize():
    from marigold_amd.util import batchsize, image_util as iu
    img = torch.randint(0, 255, (1, 3, 384, 512), dtype=torch.uint8)
    out = iu.resize_max_res(img, 768, iu.get_tv_resample_method("bilinear"))
    assert out.shape == (1, 3, 576, 768) and out.dtype == torch.uint8     # up-scales, keeps dtype
    assert iu.resize(img, (384, 512)) is img
    with pytest.raises(ValueError):
        iu.get_tv_resample_method("lanczos")
    assert iu.get_tv_resample_method("nearest") is iu.InterpolationMode.NEAREST_EXACT
    assert iu.chw2hwc(np.zeros((3, 4, 5))).shape == (4, 5, 3)
    class _Odd:
        shape = (1, 2, 3)
    with pytest.raises(TypeError):
        iu.chw2hwc(_Odd())
    col = iu.colorize_depth_maps(np.linspace(0, 1, 12).reshape(3, 4), 0, 1)
    assert col.shape == (1, 3, 3, 4)
    if not torch.cuda.is_available():
        assert batchsize.find_batch_size(10, 768, torch.bfloat16) == 1


class _FakeModule:
    device = torch.device("cpu")
    dtype = torch.bfloat16


def test_pipeline_error_and_warning_behaviour(caplog):
    import marigold_amd as M
    from marigold_amd import schedulers as S
    pipe = M.MarigoldDepthPipeline(_FakeModule(), _FakeModule(), S.DDIMScheduler(timestep_spacing="leading",
                                                                                rescale_betas_zero_snr=False),
                                   default_denoising_steps=4, default_processing_resolution=768)
    with caplog.at_level(logging.WARNING):
        pipe._check_inference_step(4)
    assert "trailing" in caplog.text and "rescale_betas_zero_snr" in caplog.text
    with pytest.raises(AssertionError):
        pipe._check_inference_step(0)
    with pytest.raises(AssertionError):
        pipe(torch.zeros(1, 3, 8, 8, dtype=torch.uint8), processing_res=-1)
    with pytest.raises(AssertionError):
        pipe(torch.zeros(1, 3, 8, 8, dtype=torch.uint8), ensemble_size=0)
    with pytest.raises(TypeError):
        pipe(np.zeros((8, 8, 3)), processing_res=0)
    with pytest.raises(AssertionError):
        pipe(torch.zeros(3, 8, 8, dtype=torch.uint8), processing_res=0)
    with pytest.raises(ValueError):
        pipe(torch.zeros(1, 3, 8, 8, dtype=torch.uint8), resample_method="lanczos")
    pipe.scheduler = object()
    with pytest.raises(RuntimeError):
        pipe._check_inference_step(1)
    npipe = M.MarigoldNormalsPipeline(_FakeModule(), _FakeModule(), S.LCMScheduler())
    with pytest.raises(RuntimeError):
        npipe._check_inference_step(4)
    dpipe = M.MarigoldDepthPipeline(_FakeModule(), _FakeModule(), S.LCMScheduler())
    with caplog.at_level(logging.WARNING):
        dpipe._check_inference_step(4)
    assert "LCMScheduler will not be supported" in caplog.text
    assert M.MarigoldPipeline is M.MarigoldDepthPipeline
    assert M.MarigoldDepthPipeline.latent_scale_factor == 0.18215


def test_ensemble_argument_validation_without_gpu():
    from marigold_amd import ensemble as ens
    x = torch.rand(3, 1, 8, 8)
    with pytest.raises(ValueError):
        ens.ensemble_depth(x, False, True)
    with pytest.raises(ValueError):
        ens.ensemble_depth(x[:, 0], True, True)
    with pytest.raises(ValueError):
        ens.ensemble_depth(x, True, True, reduction="max")
    with pytest.raises(ValueError):
        ens.ensemble_normals(torch.rand(3, 2, 8, 8))
    with pytest.raises(ValueError):
        ens.ensemble_normals(torch.rand(3, 3, 8, 8), reduction="median")
    # like the reference (marigold/util/ensemble.py:39-49) no ensemble size is refused: the shape checks are the only ones
    assert ens.MAX_ENSEMBLE_SIZE is None
    with pytest.raises(ValueError, match="Expecting 4D tensor"):
        ens.ensemble_depth(torch.rand(200, 8, 8), True, True)


def test_alignment_host_arithmetic_any_ensemble_size():
    """mg_ens_align_cost_grad (the E x E closed form of the pairwise cost, numpy's summation order included) against the numpy
    form it restates, bit for bit, beyond the 128 members the first version's fixed arrays held; and the native optimiser on an
    objective of more than 256 variables."""
    import ctypes
    from marigold_amd import ensemble as ens, _lib as L

    class _Stats:
        def __init__(self, E, g):
            x = torch.rand(E, 64, generator=g, dtype=torch.float64) + 0.1 * torch.arange(E, dtype=torch.float64)[:, None] / E
            self.m = x.mean(1).numpy()
            xc = x - x.mean(1, keepdim=True)
            self.C = (xc @ xc.T / 64).numpy()
            self.lo, self.hi = x.min(1).values.numpy(), x.max(1).values.numpy()

        def stats(self):
            return self.lo, self.hi, self.m, self.C

    g = torch.Generator().manual_seed(3)
    for E in (7, 129, 200):
        al = ens.DepthAligner(torch.zeros(E, 1, 2, 2), True, True, "median", 0.0, backend=_Stats(E, g))
        p = al.init_param() * (1.0 + 0.01 * np.random.RandomState(E).rand(2 * E))
        c1, g1 = al.cost_and_grad(p)
        c2, g2 = al.cost_and_grad_numpy(p)
        assert c1 == c2 and np.array_equal(g1, g2), E
    # the optimiser itself has no size limit either: a 300-variable convex quadratic ends at its minimum
    n = 300
    FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_double))
    w = 1.0 + np.arange(n) / n

    def fn(_u, nn, x, f, gr):
        xv = np.ctypeslib.as_array(x, (nn,))
        f[0] = float(0.5 * np.sum(w * (xv - 1.0) ** 2))
        np.ctypeslib.as_array(gr, (nn,))[:] = w * (xv - 1.0)
        return 0
    cb = FN(fn)
    x = np.zeros(n)
    fval, nit, nfev, status = ctypes.c_double(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = L.load().mg_bfgs_minimize(ctypes.cast(cb, ctypes.c_void_p), None, n, x.ctypes.data, 1e-8, 500, ctypes.byref(fval),
                                   ctypes.byref(nit), ctypes.byref(nfev), ctypes.byref(status))
    assert rc == 0 and status.value == 0 and np.abs(x - 1.0).max() < 1e-6


def test_gather_members_forced_in_a_group_of_one():
    """``force=True`` runs the collective in a one-rank group (how the RCCL path is executed on a single GPU,
    bench.py MARIGOLD_BENCH_FORCE_DIST=1); here on gloo: rooted gather and all_gather to self return the members."""
    import socket
    import torch.distributed as dist
    from marigold_amd import dist as md
    x = torch.rand(3, 1, 4, 6)
    assert md.gather_members(x, 3, (1, 4, 6), torch.device("cpu"), force=True) is x   # no process group: nothing to force
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert md.gather_members(x, 3, (1, 4, 6), torch.device("cpu")) is x
        for root in (None, 0):
            got = md.gather_members(x, 3, (1, 4, 6), torch.device("cpu"), None, root, force=True)
            assert got is not x and torch.equal(got, x)
    finally:
        dist.destroy_process_group()


def _gloo_worker(rank, world, port, E, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from marigold_amd import dist as md
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = md.shard_members(E, world, rank)
        local = torch.stack([torch.full((1, 4, 6), float(e)) for e in mine]) if mine else None
        full = md.gather_members(local, E, (1, 4, 6), torch.device("cpu"), root=None)
        rooted = md.gather_members(local, E, (1, 4, 6), torch.device("cpu"), root=1)
        ok = torch.equal(full[:, 0, 0, 0], torch.arange(E, dtype=torch.float32))
        ok &= (rooted is None) == (rank != 1)
        if rooted is not None:
            ok &= torch.equal(rooted, full)
        q.put((rank, bool(ok), mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("E", [2, 5])
def test_member_parallel_gather_gloo_world2(E):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + E
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, E, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    owned = sorted(e for _, _, m in res for e in m)
    assert owned == list(range(E))


def test_full_size_programs_validate_without_gpu():
    """The SD-v2 UNet (B=10 members, 96x96 latent, 2 DDIM steps) and the AutoencoderKL programs at
    768x768 are emitted against host buffers and every op is checked against its kernel's
    shape/alignment contract through the C ABI (mg_program_validate) - no device needed.  Also pins
    the algorithmic FLOPs per member used for the roofline."""
    from marigold_amd import opstats
    from marigold_amd.arch import UNetConfig, VAEConfig, unet_param_shapes, vae_param_shapes
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler
    ucfg, vcfg = UNetConfig(), VAEConfig()
    usd = {k: torch.zeros(s) for k, s in unet_param_shapes(ucfg).items()}
    vsd = {k: torch.zeros(s) for k, s in vae_param_shapes(vcfg).items()}
    unet = UNet2DConditionModelHIP(usd, ucfg).dry()
    unet.set_context(torch.zeros(1, 2, 1024))
    B = 10
    prog = unet.denoise_program(B, 96, 96, DDIMScheduler(), 2)
    prog.seq.validate()
    per_fwd = opstats.program_flops(prog.seq.ops[prog.n_prologue_ops:]) / 2 / B / 1e9
    # SURVEY: 2137.7 GF with the full cross-attention; the 2-token collapse removes ~51 GF and the sub-pixel form of
    # the three up-sampling convolutions (4/9 of their MACs) another 84.9 GF: 2001.5 GF executed per member and forward
    assert abs(per_fwd - 2001.5) < 3, per_fwd
    # the launch forms of round 5, as the engine routes them at full size
    from marigold_amd import _lib as L
    labelled = list(zip(prog.seq.ops, prog.seq.labels))
    fwd = labelled[prog.n_prologue_ops:prog.n_prologue_ops + prog.n_fwd_ops]
    folded = [op for op, lab in fwd if lab.endswith("conv2+conv_shortcut")]   # conv_shortcut as extra K of conv2 (implicit GEMM levels)
    # (round 6: at ten members the 96 x 96 level's norms are separate passes and its plain convolutions run on the hand-placed GEMM
    # tile - engine.fuse_norm_into_conv - so up_blocks.3's three shortcuts fold as well: 14 folded launches, no shortcut launch left)
    assert len(folded) == 14 and all(op.kind == L.OP_IGEMM and op.p[12] and op.i[32] % 64 == 0 and op.i[7] == 9 for op in folded)
    assert sum(bool(op.p[13]) for op in folded) == 12         # the up blocks' [hidden | skip] pairs, never concatenated
    assert [lab for _, lab in fwd if lab.endswith(".conv_shortcut")] == []
    assert any(op.kind == L.OP_GN_STATS and op.p[6] and op.i[9] > 0 for op, _ in fwd)      # a skip concat's statistics: one launch
    assert not any(lab.endswith((".stats0", ".stats1")) for _, lab in fwd)
    assert [lab for _, lab in labelled[:prog.n_prologue_ops]].count("resnets.time_emb_proj") == 1
    # the UNet's 96^2 head stays on the pass + GEMM pair, the decoder's 768^2 head is one MG_OP_CONV3X3_HEAD launch
    assert [lab for _, lab in fwd if "conv_out" in lab or "conv_norm_out" in lab] == ["conv_norm_out.stats", "conv_norm_out.apply", "conv_out", "conv_out.post+scheduler.step"]
    vae = AutoencoderKLHIP(vsd, vcfg).dry()
    seq, _, _ = vae._program("encode", 1, 768, 768)
    seq.validate()
    assert abs(opstats.program_flops(seq.ops) / 1e9 - 2609.1) < 5
    seq, _, _ = vae._program("decode", 2, 96, 96, 1)
    seq.validate()
    heads = [op for op in seq.ops if op.kind == L.OP_CONV3X3_HEAD]
    assert len(heads) == 1 and (heads[0].i[1], heads[0].i[2], heads[0].i[3], heads[0].i[4]) == (768, 768, 128, 3) and heads[0].p[1]
    # 5754.3 GF as published; the three up-sampling convolutions in sub-pixel form execute 869.8 GF less
    assert abs(opstats.program_flops(seq.ops) / 2 / 1e9 - 4884.5) < 5
    # the IID family at full size (appearance: 2 modalities -> 12 in / 8 out latent channels; lighting: 3 -> 16 / 12):
    # conv_in / conv_out widths differ, the modalities ride in the batch of ONE decode program
    for n_targets in (2, 3):
        icfg = UNetConfig(in_channels=4 + 4 * n_targets, out_channels=4 * n_targets)
        iunet = UNet2DConditionModelHIP({k: torch.zeros(sh) for k, sh in unet_param_shapes(icfg).items()}, icfg).dry()
        iunet.set_context(torch.zeros(1, 2, 1024))
        iprog = iunet.denoise_program(2, 96, 96, DDIMScheduler(), 1)
        iprog.seq.validate()
        assert tuple(iprog.x.shape) == (2, 4 * n_targets, 96, 96)
        seq, _, _ = vae._program("decode", 2 * n_targets, 96, 96, 3)   # post = MG_POST_UNIT
        seq.validate()
    # odd sizes (C1: 384x512 input up-scaled to 576x768 -> 72x96 latent) and a contract violation
    unet.denoise_program(1, 72, 96, DDIMScheduler(timestep_spacing="leading", rescale_betas_zero_snr=False), 1).seq.validate()
    from marigold_amd import _lib as L, ops as O
    bad = O.OpSeq("bad")
    buf = torch.zeros(64, dtype=torch.uint8)
    bad.add(O.linear(buf, buf, buf, M=4, K=40, N=8), "K not a multiple of 64")
    with pytest.raises(L.MarigoldHipError, match="multiple of 64"):
        bad.validate()
    # the fused cross-attention epilogue takes exactly 64 score columns and a second stage of whole 32-channel blocks
    buf = torch.zeros(1 << 16, dtype=torch.uint8)
    for kw, what in ((dict(N=128, c2=320), "128 score columns"), (dict(N=64, c2=40), "c2 not a multiple of 32"),
                     (dict(N=64, c2=320, out2=False), "no second-stage weights")):
        kw = dict(kw)
        w2 = buf if kw.pop("out2", True) else None
        bad = O.OpSeq("bad")
        bad.add(O.linear(buf, buf, buf, M=128, K=64, epi=L.EPI_XATTN2, sm_scale=0.125, sm_cols=10, out2=w2, ldo=320, **kw), what)
        with pytest.raises(L.MarigoldHipError, match="fused cross-attention"):
            bad.validate()
    ok = O.OpSeq("ok")
    ok.add(O.linear(buf, buf, buf, M=128, K=64, N=64, epi=L.EPI_XATTN2, sm_scale=0.125, sm_cols=10, out2=buf, c2=320, ldo=320), "xattn2")
    ok.validate()


def test_fast_division_constants():
    """csrc/common.h: mg_make_fastdiv / fdiv - floor(x / d) for x < 2^31 as umulhi(2x, m) >> L with L = ceil(log2 d),
    m = ceil(2^(31+L) / d).  Restated in integer arithmetic and checked on the divisors the launchers form (tile counts,
    rows per image, widths, patch widths) and on the edges of the claim."""
    def make(d):
        L_ = 0
        while (1 << L_) < d:
            L_ += 1
        m = ((1 << (31 + L_)) + d - 1) // d
        assert m < (1 << 32)
        return m, L_

    def fdiv(x, m, L_):
        return ((((x << 1) & 0xFFFFFFFF) * m) >> 32) >> L_

    rng = np.random.default_rng(0)
    divisors = [1, 2, 3, 5, 7, 10, 17, 18, 20, 36, 40, 45, 72, 90, 96, 144, 360, 576, 720, 2304, 9216, 92160, 589824,
                (1 << 31) - 1, 1 << 30, (1 << 30) + 1, 3 * (1 << 29)] + [int(v) for v in rng.integers(1, 1 << 31, 200)]
    for d in divisors:
        m, L_ = make(d)
        xs = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d,
              ((1 << 31) - 1) // d * d - 1] + [int(v) for v in rng.integers(0, 1 << 31, 300)]
        for x in xs:
            if 0 <= x < (1 << 31):
                assert fdiv(x, m, L_) == x // d, (x, d)


def test_alignment_objective_native_equals_numpy_bit_for_bit():
    """``mg_ens_align_cost_grad`` (one C call per BFGS evaluation) restates the numpy form of the pairwise-RMSE objective
    operation for operation, numpy's pairwise summation included: same bits for every ensemble size the kernels accept,
    affine / scale-only, median / mean, coincident members (zero distances) and generic ones."""
    from marigold_amd import ensemble as ens

    class Backend:
        def __init__(self, E, seed):
            rng = np.random.default_rng(seed)
            self.E, self.seed, self.dm, self.dx = E, seed, rng.random(E), rng.random(E)

        def stats(self):
            rng = np.random.default_rng(self.seed + 1)
            A = rng.standard_normal((self.E, 40))
            return np.zeros(self.E), np.ones(self.E), rng.random(self.E), A @ A.T / 40

        def regulariser(self, s32, t32):
            return 0.013, 0.97, self.dm, self.dx

    n = 0
    for E in (1, 2, 3, 7, 8, 9, 10, 16, 17, 31, 32):
        for affine in (True, False):
            for red in ("median", "mean"):
                al = ens.DepthAligner(torch.zeros(E, 1, 4, 4), True, affine, red, 0.02, backend=Backend(E, E))
                rng = np.random.default_rng(E)
                for it in range(6):
                    p = al.init_param() + rng.standard_normal(2 * E if affine else E) * 0.05
                    if it % 3 == 0:
                        p[:] = al.init_param()
                    f1, g1 = al.cost_and_grad(p)
                    f0, g0 = al.cost_and_grad_numpy(p)
                    assert f0 == f1 and np.array_equal(g0, g1), (E, affine, red, it)
                    n += 1
    assert n == 264


# ---- depth-ensembling host logic vs the reference's own outputs (CPU backend for the pixel passes)

def _align_cpu(x, **kw):
    import scipy.optimize
    from marigold_amd import ensemble as ens
    from tests.cpu_backend import TorchStatsBackend
    be = TorchStatsBackend(x, 0, True)
    al = ens.DepthAligner(x.float(), True, True, "median", 0.02, backend=be)
    p0 = al.init_param()
    res = scipy.optimize.minimize(al.reference_fd_objective, p0, jac=True, method="BFGS", tol=1e-6,
                                  options={"maxiter": 50})
    return al, p0, res


@pytest.mark.parametrize("name,tol_max,tol_mean", [("d_real_e10", 4e-2, 8e-3), ("d_real_e4", 8e-3, 2e-3),
                                                   ("d_e4", 1e-2, 3e-3), ("d_e10", 6e-2, 1.2e-2),
                                                   ("d_e3", 9e-2, 2.5e-2)])
def test_depth_alignment_reproduces_reference_behaviour(golden_dir, name, tol_max, tol_mean):
    """(1) scales stay EXACTLY at the reference's init (its finite differences cannot move them);
    (2) our shifts are at least as good as the reference's under the ORACLE's cost function;
    (3) the ensembled map is close to the reference's output.  It cannot be identical: the reference
    stops its shift optimisation early, where fp32 summation noise swamps its finite differences
    (e.g. d_e3: it returns init_param unchanged), so the bound is the gap between ITS stopping point
    and the optimum it was descending to; on members that agree like real predictions it is small."""
    from oracle import ensemble as oens
    gold = np.load(os.path.join(golden_dir, "ensemble_ref.npz"))
    x = torch.from_numpy(gold[f"{name}_in"])
    E = x.shape[0]
    al, p0, res = _align_cpu(x)
    p = res.x
    assert np.array_equal(p[:E], p0[:E]), "scales must not move (reference FD step dies in the fp32 cast)"
    _, _, pref = oens.ensemble_depth(x, True, True, return_param=True)
    np.testing.assert_allclose(pref[:E], p0[:E], rtol=0, atol=1e-6)   # ... and the reference's do not either
    c_ref = oens.depth_cost(pref, x.float(), True, True, "median", 0.02)
    c_ours = oens.depth_cost(p, x.float(), True, True, "median", 0.02)
    assert abs(al.cost(p) - c_ours) < 2e-5, "closed-form cost == the reference's pairwise loop"
    assert c_ours <= c_ref + 1e-5, (c_ours, c_ref)
    a = oens.depth_align(x, p, True, True)
    d, _ = oens.depth_reduce(a, "median", False)
    d = (d - d.min()) / (d.max() - d.min()).clamp(min=1e-6)
    diff = (d - torch.from_numpy(gold[f"{name}_out"])).abs()
    print(f"[parity] {name}: cost ours {c_ours:.5f} <= ref {c_ref:.5f}; {al.n_eval} evals; "
          f"|out-ref| max {float(diff.max()):.4f} mean {float(diff.mean()):.5f}")
    assert float(diff.max()) < tol_max and float(diff.mean()) < tol_mean


def test_depth_alignment_gradient_and_fd_survival(golden_dir):
    from marigold_amd import ensemble as ens
    gold = np.load(os.path.join(golden_dir, "ensemble_ref.npz"))
    x = torch.from_numpy(gold["d_real_e4_in"])
    al, p0, _ = _align_cpu(x)
    p = ens._q32(p0 + 0.01)
    f0, g = al.cost_and_grad(p)
    for k in range(len(p)):      # central differences on fp32-representable points
        h = 2.0 ** -12
        pp, pm = p.copy(), p.copy()
        pp[k] += h
        pm[k] -= h
        fd = (al.cost(pp) - al.cost(pm)) / (2 * h)
        assert abs(fd - g[k]) < 2e-3 * max(1.0, abs(g[k])), (k, fd, g[k])
    k = ens.fd_survival(np.array([1.25, 3.0, 0.2, 0.01, -0.01, 0.0]))
    assert k[0] == 0 and k[1] == 0 and abs(k[2] - 1) < 1e-6 and abs(k[3] - 1) < 0.07 and abs(k[5] - 1) < 1e-6


def test_map_images_host_logic():
    """pipeline.map_images without a GPU: one map at a time on the caller's thread, inputs consumed lazily and zipped with their
    generators, argument checks made at the call (the several-lanes form is a GPU test:
    tests/test_gpu_pipeline.py::test_map_images_maps_in_flight_bit_identical)."""
    from marigold_amd.pipeline import _MarigoldPipelineBase

    class P(_MarigoldPipelineBase):
        device = torch.device("cpu")

        def __init__(self):
            self._member_parallel = False
            self.seen = []

        def __call__(self, image, **kw):
            self.seen.append((image, kw.get("generator"), kw.get("ensemble_size")))
            return image * 2

    p = P()
    pulled = []

    def src():
        for k in range(4):
            pulled.append(k)
            yield k
    it = p.map_images(src(), in_flight=3, generators=iter("abcd"), ensemble_size=5)
    assert pulled == [] and p.seen == []          # nothing runs before the first output is asked for
    assert next(it) == 0 and pulled == [0]        # ... and no further than the map that is asked for
    assert list(it) == [2, 4, 6]
    assert p.seen == [(0, "a", 5), (1, "b", 5), (2, "c", 5), (3, "d", 5)]
    assert list(p.map_images([1, 2], in_flight=1, generator="g")) == [2, 4]     # a shared generator is fine one map at a time
    # default lane count: by the members this GPU runs per map (three lanes up to eight members, two beyond)
    assert [p.maps_in_flight_for(e) for e in (1, 5, 8, 9, 10)] == [3, 3, 3, 2, 2]
    with pytest.raises(ValueError):
        p.map_images([1, 2, 3], generators=["a"])
    with pytest.raises(ValueError):
        p.map_images([1], in_flight=0)
    with pytest.raises(ValueError):
        list(p.map_images(iter([1, 2, 3]), generators=iter(["a"])))


def test_cli_flags_and_output_formats(tmp_path):
    """script/{depth,normals}/run.py: the reference's flags, folders, file names and encodings
    (script/depth/run.py:54-135, 165-171, 270-292), driven with a stand-in pipeline object."""
    from PIL import Image
    from marigold_amd import cli
    from marigold_amd.pipeline import MarigoldDepthOutput, MarigoldNormalsOutput
    a = cli.build_parser("depth").parse_args(["--input_rgb_dir", "i", "--output_dir", "o", "--fp16", "--seed", "3"])
    assert a.checkpoint == "prs-eth/marigold-depth-v1-1" and a.half_precision and a.ensemble_size == 1
    assert a.color_map == "Spectral" and a.resample_method == "bilinear" and a.batch_size == 0
    assert not hasattr(cli.build_parser("normals").parse_args(["--input_rgb_dir", "i", "--output_dir", "o"]), "color_map")
    inp = tmp_path / "in"
    inp.mkdir()
    Image.fromarray(np.zeros((6, 8, 3), np.uint8)).save(inp / "b.png")
    Image.fromarray(np.zeros((6, 8, 3), np.uint8)).save(inp / "a.JPG")
    (inp / "notes.txt").write_text("x")
    calls = []

    class FakeDepth:
        scale_invariant = shift_invariant = True
        default_denoising_steps, default_processing_resolution = 4, 768

        def __call__(self, image, **kw):
            calls.append(kw)
            d = np.linspace(0, 1, 48, dtype=np.float32).reshape(6, 8)
            return MarigoldDepthOutput(d, Image.fromarray(np.zeros((6, 8, 3), np.uint8)), None)

    out = tmp_path / "out"
    assert cli.main("depth", ["--input_rgb_dir", str(inp), "--output_dir", str(out), "--ensemble_size", "2",
                              "--output_processing_res"], pipeline=FakeDepth()) == 0
    assert len(calls) == 2 and calls[0]["ensemble_size"] == 2 and calls[0]["match_input_res"] is False
    assert calls[0]["color_map"] == "Spectral" and calls[0]["generator"] is None
    assert np.load(out / "depth_npy" / "a_depth.npy").dtype == np.float32
    bw = np.array(Image.open(out / "depth_bw" / "b_depth.png"))
    assert bw.dtype == np.uint16 and bw.max() == 65535 and bw.min() == 0
    assert (out / "depth_colored" / "a_depth_colored.png").exists()

    class FakeNormals:
        default_denoising_steps, default_processing_resolution = 4, 768

        def __call__(self, image, **kw):
            assert "color_map" not in kw
            n = np.zeros((3, 6, 8), np.float32)
            n[2] = 1
            return MarigoldNormalsOutput(n, Image.fromarray(np.zeros((6, 8, 3), np.uint8)), None)

    assert cli.main("normals", ["--input_rgb_dir", str(inp), "--output_dir", str(out)], pipeline=FakeNormals()) == 0
    assert np.load(out / "normals_npy" / "b_normals.npy").shape == (3, 6, 8)
    assert (out / "normals_vis" / "a_normals.png").exists()

    from marigold_amd.pipeline import MarigoldIIDOutput
    props = {"target_names": ["albedo", "shading", "residual"], "albedo": {"prediction_space": "srgb"},
             "shading": {"prediction_space": "linear", "up_to_scale": True},
             "residual": {"prediction_space": "linear", "up_to_scale": True}}

    class FakeIID:
        default_denoising_steps, default_processing_resolution = 4, 768
        target_names = props["target_names"]

        def __call__(self, image, **kw):
            o = MarigoldIIDOutput(self.target_names)
            for n in self.target_names:
                o.fill_entry(n, torch.rand(1, 3, 6, 8), None, props)
            return o

    assert cli.build_parser("iid").parse_args(["--input_rgb_dir", "i", "--output_dir", "o"]).checkpoint == \
        "prs-eth/marigold-iid-appearance-v1-1"
    assert cli.main("iid", ["--input_rgb_dir", str(inp), "--output_dir", str(out), "--checkpoint",
                            "prs-eth/marigold-iid-lighting-v1-1"], pipeline=FakeIID()) == 0
    assert np.load(out / "iid_lighting_npy" / "a_shading.npy").shape == (6, 8, 3)   # chw2hwc like script/iid/run.py:263
    assert Image.open(out / "iid_lighting_vis" / "b_residual.png").size == (8, 6)
    assert cli.main("iid", ["--input_rgb_dir", str(inp), "--output_dir", str(out)], pipeline=FakeIID()) == 0
    assert (out / "iid_appearance_npy" / "b_albedo.npy").exists()
    assert cli.main("depth", ["--input_rgb_dir", str(tmp_path / "out"), "--output_dir", str(out)],
                    pipeline=FakeDepth()) == 1   # no images -> exit code 1 like the reference


def test_from_pretrained_without_checkpoint_fails_clearly():
    import marigold_amd as M
    with pytest.raises(FileNotFoundError, match="neither a local folder nor in the local Hugging Face cache"):
        M.MarigoldDepthPipeline.from_pretrained("prs-eth/marigold-depth-v1-1")


def _tiny_checkpoint(tmp_path, with_text_encoder):
    """A checkpoint folder in the diffusers layout with the tiny architecture and the FULL published config key
    sets; optionally with a real (tiny, random) CLIP text encoder + tokenizer instead of the precomputed embedding."""
    import json
    from marigold_amd import checkpoint as ck, synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE
    from marigold_amd.schedulers import DDIMScheduler
    path = str(tmp_path / "ckpt")
    ck.save_synthetic_checkpoint(path, "MarigoldDepthPipeline", syn.synthetic_unet_state_dict(TINY_UNET),
                                 syn.synthetic_vae_state_dict(TINY_VAE), TINY_UNET, TINY_VAE, DDIMScheduler(),
                                 syn.synthetic_text_embedding(TINY_UNET.cross_attention_dim), scale_invariant=True,
                                 shift_invariant=True, default_denoising_steps=4, default_processing_resolution=768)
    if with_text_encoder:
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
        os.remove(os.path.join(path, "empty_text_embed.safetensors"))
        vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1, "a</w>": 2, "b</w>": 3}
        json.dump(vocab, open(os.path.join(path, "vocab.json"), "w"))
        open(os.path.join(path, "merges.txt"), "w").write("#version: 0.2\n")
        tok = CLIPTokenizer(os.path.join(path, "vocab.json"), os.path.join(path, "merges.txt"), model_max_length=77)
        tok.save_pretrained(os.path.join(path, "tokenizer"))
        torch.manual_seed(0)
        cfg = CLIPTextConfig(vocab_size=4, hidden_size=TINY_UNET.cross_attention_dim, intermediate_size=128, num_hidden_layers=2,
                             num_attention_heads=2, max_position_embeddings=77, bos_token_id=0, eos_token_id=1, pad_token_id=1)
        CLIPTextModel(cfg).eval().save_pretrained(os.path.join(path, "text_encoder"))
    return path


def test_encode_empty_text_with_a_real_text_encoder(tmp_path):
    """A9 (marigold_depth_pipeline.py:381-394): the checkpoint's CLIP text encoder + tokenizer are loaded, "" is
    tokenised with padding="do_not_pad" -> exactly the two tokens [bos, eos], and the [1, 2, D] embedding becomes the
    UNet's 2-token context (the collapse of every cross-attention rests on that length)."""
    import marigold_amd as M
    path = _tiny_checkpoint(tmp_path, with_text_encoder=True)
    pipe = M.MarigoldDepthPipeline.from_pretrained(path)
    assert pipe.empty_text_embed is None and pipe.text_encoder is not None and pipe.tokenizer is not None
    ids = pipe.tokenizer("", padding="do_not_pad", max_length=pipe.tokenizer.model_max_length, truncation=True,
                         return_tensors="pt").input_ids
    assert ids.tolist() == [[0, 1]]
    pipe.encode_empty_text()
    emb = pipe.empty_text_embed
    assert tuple(emb.shape) == (1, 2, 64) and torch.isfinite(emb.float()).all()
    with torch.no_grad():
        want = pipe.text_encoder(ids)[0]
    torch.testing.assert_close(emb.float(), want.to(emb.dtype).float())
    pipe.unet.set_context(emb)            # accepted as the 2-token context; any other length is refused
    with pytest.raises(ValueError, match="2-token context"):
        pipe.unet.set_context(torch.zeros(1, 77, 64))


def test_tuning_table_is_well_formed_and_applied():
    """marigold_amd/tuning/gfx950.json (the measured tile / split-K choices of MG_OP_IGEMM): every entry names a tile the library
    can dispatch and a split count it accepts, keys parse as the 13 integers ``tuning.key_of`` writes, every entry was worth at
    least the 6 % its sweep demanded; ``engine.Builder.add`` applies an entry to a matching op and leaves a caller's own choice
    alone; a full-size program built with the table validates."""
    import re
    from marigold_amd import _lib as L, ops as O, tuning
    db = tuning.load()
    assert len(db) >= 100
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marigold_amd", "csrc", "igemm2.hip")).read()
    tiles = {int(v) for v in re.findall(r"\n    case (\d+):", src)}
    assert {23, 32, 35, 36, 46, 62, 72, 73} <= tiles
    for key, (variant, splits, t_auto, t_best, where) in db.items():
        parts = key.split(",")
        if len(parts) == 14:   # conv2 + conv_shortcut in one launch: the folded 1x1 convolution's channel count
            assert re.fullmatch(r"x\d+", parts[13]) and int(parts[13][1:]) % 64 == 0 and parts[3] == "9", key
            parts = parts[:13]
        assert len(parts) == 13 and all(re.fullmatch(r"-?\d+", x) for x in parts), key
        assert variant in tiles and 1 <= splits <= 24, (key, variant, splits)
        assert t_best <= 0.94 * t_auto + 0.11, (key, t_auto, t_best)   # (both are stored rounded to 0.1 us)
        if int(parts[5]) != L.EPI_BF16 or int(parts[9]) or int(parts[10]) or int(parts[6]) or int(parts[7]) > 1:
            assert splits == 1, f"{key}: split-K only for the plain bf16 epilogue"   # row statistics / folded LayerNorm / GEGLU / V^T
    # an op matching a table entry gets the entry; variant / splits given by the caller are left alone
    def usable(k):
        p_ = k.split(",")
        return (len(p_) == 13 and (p_[3], p_[4], p_[5], p_[6], p_[7], p_[9], p_[10], p_[11]) == ("9", "1", "0", "0", "1", "0", "0", "0") and
                int(int(p_[0]) ** 0.5) ** 2 == int(p_[0]))
    key, (variant, splits, *_rest) = next((k, v) for k, v in db.items() if usable(k))
    parts = key.split(",")
    M_, N_, K_ = (int(x) for x in parts[:3])
    HW = int(M_ ** 0.5)
    a = torch.zeros(1)
    kw = dict(B=1, H=HW, W=HW, Cin=K_ // 9, Ho=HW, Wo=HW, N=N_, taps=9, stride=1, pad=1, bias=a, residual=a if parts[8] == "1" else None,
              rowvec=a if parts[12] == "1" else None)
    op = O.igemm(a, a, a, **kw)
    assert tuning.key_of(op) == key
    assert (tuning.apply(op).i[19], op.i[31]) == (variant, splits)
    op2 = O.igemm(a, a, a, variant=23, **kw)
    assert (tuning.apply(op2).i[19], op2.i[31]) == (23, 0)
    # a folded launch has its own entries (more K than the plain convolution of the same shape)
    fkey, (fv, fs, *_r) = next((k, v) for k, v in db.items() if k.count(",") == 13 and int(int(k.split(",")[0]) ** 0.5) ** 2 == int(k.split(",")[0]))
    fp = fkey.split(",")
    HW, cx = int(int(fp[0]) ** 0.5), int(fp[13][1:])
    opf = O.igemm(a, a, a, B=1, H=HW, W=HW, Cin=int(fp[2]) // 9, Ho=HW, Wo=HW, N=int(fp[1]), taps=9, stride=1, pad=1, bias=a,
                  residual=a if fp[8] == "1" else None, rowvec=a if fp[12] == "1" else None, fold=(a, None, cx, cx))
    assert tuning.key_of(opf) == fkey and (tuning.apply(opf).i[19], opf.i[31]) == (fv, fs)


def test_folded_shortcut_and_stacked_time_projection_weights():
    """Host arithmetic of two weight layouts of round 5 (engine.WeightStore), against torch on the CPU: (a) ``conv3x3_fold`` - rows
    [9 Cin | Cx] and the summed bias reproduce conv2(h) + conv_shortcut(x) (diffusers ResnetBlock2D's output sum) as ONE GEMM over
    [im2col(h) | x]; (b) ``time_emb_proj_all`` - the ResNet blocks' projections stacked along N give each block's own Linear."""
    import torch.nn.functional as F
    from marigold_amd import engine as E
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cx, Cout = 2, 6, 5, 64, 128, 64
    sd = {"r.conv2.weight": torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05, "r.conv2.bias": torch.randn(Cout, generator=g),
          "r.conv_shortcut.weight": torch.randn(Cout, Cx, 1, 1, generator=g) * 0.05, "r.conv_shortcut.bias": torch.randn(Cout, generator=g),
          "a.time_emb_proj.weight": torch.randn(64, 32, generator=g), "a.time_emb_proj.bias": torch.randn(64, generator=g),
          "b.time_emb_proj.weight": torch.randn(128, 32, generator=g), "b.time_emb_proj.bias": torch.randn(128, generator=g)}
    ws = E.WeightStore(sd, torch.device("cpu"))
    wf, bf = ws.conv3x3_fold("r.conv2", "r.conv_shortcut")
    assert wf.dtype == torch.bfloat16 and tuple(wf.shape) == (Cout, 9 * Cin + Cx) and bf.dtype == torch.float32
    h = torch.randn(B, Cin, H, W, generator=g)
    x = torch.randn(B, Cx, H, W, generator=g)
    # im2col in the kernel's K order: k = (ky * 3 + kx) * Cin + c, then the shortcut's channels at the centre pixel
    cols = F.unfold(h, 3, padding=1).reshape(B, Cin, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * Cin)
    a = torch.cat([cols, x.permute(0, 2, 3, 1).reshape(B * H * W, Cx)], dim=1)
    got = (a @ wf.float().t() + bf).reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    wq = lambda k: sd[k].to(torch.bfloat16).float()
    ref = F.conv2d(h, wq("r.conv2.weight"), sd["r.conv2.bias"], padding=1) + F.conv2d(x, wq("r.conv_shortcut.weight"), sd["r.conv_shortcut.bias"])
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    wall, ball = ws.time_emb_proj_all(["a", "b"])
    e = torch.randn(3, 32, generator=g)
    out = F.silu(e) @ wall.t() + ball
    torch.testing.assert_close(out[:, :64], F.linear(F.silu(e), sd["a.time_emb_proj.weight"], sd["a.time_emb_proj.bias"]))
    torch.testing.assert_close(out[:, 64:], F.linear(F.silu(e), sd["b.time_emb_proj.weight"], sd["b.time_emb_proj.bias"]))


def test_opstats_of_the_round5_launch_forms():
    """The algorithmic FLOPs / bytes bench.py's roofline uses (marigold_amd/opstats.py) for the launch forms added in round 5: a
    folded launch = the 3x3 convolution + the 1x1 convolution it carries (its input and weights read once, no residual round
    trip); the output head = one read of the input; a two-source statistics launch = both sources' bytes."""
    from marigold_amd import _lib as L, ops as O, opstats
    a = torch.zeros(1)
    B, HW, Cin, Cx, N = 2, 24, 128, 192, 256
    kw = dict(B=B, H=HW, W=HW, Cin=Cin, Ho=HW, Wo=HW, N=N, taps=9, stride=1, pad=1, bias=a)
    M = B * HW * HW
    c0, f0, b0 = opstats.op_cost(O.igemm(a, a, a, **kw))
    c1, f1, b1 = opstats.op_cost(O.igemm(a, a, a, fold=(a, a, 128, Cx), **kw))
    assert c0 == c1 == "igemm_mfma" and f0 == 2 * M * N * 9 * Cin
    assert f1 - f0 == 2 * M * N * Cx and b1 - b0 == (M + N) * Cx * 2
    ch, fh, bh = opstats.op_cost(O.conv3x3_head(a, a, a, a, a, B=B, H=HW, W=HW, C=Cin, Cout=3, ldo=8))
    assert ch == "boundary_conv" and fh == 2 * M * 3 * 9 * Cin and bh == M * (Cin * 2 + 3 * 4)
    _, _, bs1 = opstats.op_cost(O.gn_stats(a, a, B=B, HW=HW * HW, C=Cin, chunks=4, groups=32))
    _, _, bs2 = opstats.op_cost(O.gn_stats(a, a, B=B, HW=HW * HW, C=Cin, chunks=4, groups=32, Ctot=Cin + Cx, x1=a, C1=Cx))
    assert bs1 == M * Cin * 2 and bs2 == M * (Cin + Cx) * 2


def test_model_image_round_trip_without_gpu(tmp_path):
    """SURVEY section 8(b)'s module-level C entry points: ``export_model_image`` compiles the three native programs of a fixed
    shape (VAE encode, the whole T-step denoising loop, VAE decode) + their kernel-ready weights + a memory plan into one file;
    ``mg_model_load(path, device=-1)`` parses it, lays the buffers out and patches every pointer, and ``mg_model_validate`` runs
    every op's contract check - all without a GPU (tests/test_gpu_pipeline.py runs the same image on the device)."""
    import struct
    import marigold_amd as M
    from marigold_amd import image, _lib as L
    from marigold_amd.arch import TINY_UNET, TINY_VAE
    from marigold_amd.schedulers import LCMScheduler
    pipe = M.build_synthetic_pipeline("depth", TINY_UNET, TINY_VAE, default_processing_resolution=0, default_denoising_steps=2)
    path = str(tmp_path / "tiny.mgimg")
    info = image.export_model_image(pipe, path, ensemble_size=3, height=64, width=128)
    assert info["latent_hw"] == (8, 16) and info["ops"]["denoise"] > 100 and info["step_noises"] == 0
    m = image.ModelImage(path, device=-1)
    assert (m.B, m.H, m.W, m.h, m.w, m.steps, m.pred_channels, m.Hout, m.Wout) == (3, 64, 128, 8, 16, 2, 1, 64, 128)
    assert L.load().mg_model_device_bytes(m.handle) >= info["data_bytes"] + info["zero_bytes"] + info["scratch_bytes"]
    m.validate()
    with pytest.raises(L.MarigoldHipError, match="host-only"):   # a host-only model cannot run
        L.check(L.load().mg_model_vae_encode(m.handle, 1, 1, None), "mg_model_vae_encode")
    m.close()
    # the LCM scheduler draws noise in its non-final steps: the image carries one slot per draw
    lcm = M.build_synthetic_pipeline("depth", TINY_UNET, TINY_VAE, scheduler=LCMScheduler(), default_processing_resolution=0)
    info2 = image.export_model_image(lcm, str(tmp_path / "lcm.mgimg"), ensemble_size=1, height=64, width=64, denoising_steps=4)
    assert info2["step_noises"] == 3
    m2 = image.ModelImage(str(tmp_path / "lcm.mgimg"), device=-1)
    assert m2.n_noise == 3
    m2.validate()
    m2.close()
    # a file that is not an image, a truncated one, an image of another ABI: refused with a message
    bad = str(tmp_path / "bad.mgimg")
    open(bad, "wb").write(b"not an image" * 20)
    with pytest.raises(L.MarigoldHipError, match="not a model image"):
        image.ModelImage(bad, device=-1)
    raw = open(path, "rb").read()
    open(bad, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(L.MarigoldHipError, match="short read|corrupt image"):   # (round 6: the buffer table is checked against the file size first)
        image.ModelImage(bad, device=-1)
    hdr = bytearray(raw[:88])
    struct.pack_into("<I", hdr, 12, 3)   # the ABI field
    open(bad, "wb").write(bytes(hdr) + raw[88:])
    with pytest.raises(L.MarigoldHipError, match="ABI 3"):
        image.ModelImage(bad, device=-1)


def test_noise_draws_follow_the_loaded_dtype(tmp_path):
    """A4 (marigold_depth_pipeline.py:430-435): the reference draws the initial latents (and the LCM per-step noise) with
    ``dtype=self.dtype`` - fp32 for the default load, 16-bit draws for ``from_pretrained(torch_dtype=torch.float16 |
    torch.bfloat16)`` (script/depth/run.py:203-214), a DIFFERENT random stream.  ``noise_dtype`` follows the loader's
    ``torch_dtype``; the draws reach the engine widened to fp32."""
    import marigold_amd as M
    path = _tiny_checkpoint(tmp_path, with_text_encoder=False)
    shape = (3, 4, 8, 16)
    draws = {}
    for dt in (None, torch.float32, torch.float16, torch.bfloat16):
        pipe = M.MarigoldDepthPipeline.from_pretrained(path, torch_dtype=dt)
        want_dt = torch.float32 if dt in (None, torch.float32) else dt
        assert pipe.noise_dtype is want_dt
        got = pipe._randn(shape, torch.Generator().manual_seed(5))
        want = torch.randn(shape, dtype=want_dt, generator=torch.Generator().manual_seed(5))
        assert got.dtype == torch.float32 and torch.equal(got, want.float())
        draws[want_dt] = got
    assert not torch.equal(draws[torch.float32], draws[torch.bfloat16])   # not the fp32 stream rounded: another stream
    with pytest.raises(ValueError, match="torch_dtype"):
        M.MarigoldDepthPipeline.from_pretrained(path, torch_dtype=torch.float64)


def test_checkpoint_configs_are_validated_strictly(tmp_path):
    """Every field of unet/config.json, vae/config.json and scheduler_config.json is checked: the published SD-v2 key
    sets load, any value the engine does not implement (or any unknown field) raises instead of being ignored."""
    import json
    import marigold_amd as M
    from marigold_amd import config_check as CC
    # the published full-size configs pass as they are
    assert CC.unet_config_from_json(CC.SD2_UNET_CONFIG).block_out_channels == (320, 640, 1280, 1280)
    assert CC.vae_config_from_json(CC.SD2_VAE_CONFIG).latent_channels == 4
    path = _tiny_checkpoint(tmp_path, with_text_encoder=False)
    M.MarigoldDepthPipeline.from_pretrained(path)                      # full key sets, tiny sizes: loads

    def reload_with(sub, fname, **changes):
        f = os.path.join(path, sub, fname)
        orig = json.load(open(f))
        json.dump(dict(orig, **changes), open(f, "w"))
        try:
            M.MarigoldDepthPipeline.from_pretrained(path)
        finally:
            json.dump(orig, open(f, "w"))

    bad_unet = [dict(use_linear_projection=False), dict(attention_head_dim=8), dict(norm_eps=1e-6), dict(act_fn="gelu"),
                dict(only_cross_attention=True), dict(num_attention_heads=8), dict(class_embed_type="timestep"),
                dict(resnet_time_scale_shift="scale_shift"), dict(transformer_layers_per_block=2), dict(layers_per_block=3),
                dict(attention_head_dim=[2, 2, 2, 2]), dict(some_future_field=1), dict(time_embedding_type="fourier"),
                dict(dual_cross_attention=True), dict(mid_block_type="UNetMidBlock2D"), dict(addition_embed_type="text")]
    for ch in bad_unet:
        with pytest.raises(CC.UnsupportedConfigError, match=next(iter(ch))):
            reload_with("unet", "config.json", **ch)
    for ch in (dict(scaling_factor=0.13025), dict(act_fn="relu"), dict(latent_channels=16), dict(use_quant_conv=False),
               dict(mid_block_add_attention=False), dict(new_vae_field=True)):
        with pytest.raises(CC.UnsupportedConfigError, match=next(iter(ch))):
            reload_with("vae", "config.json", **ch)
    for ch in (dict(thresholding=True), dict(trained_betas=[0.1]), dict(unknown_sched_field=3)):
        with pytest.raises(CC.UnsupportedConfigError, match=next(iter(ch))):
            reload_with("scheduler", "scheduler_config.json", **ch)
    with pytest.raises(ValueError, match="clip_sample"):
        reload_with("scheduler", "scheduler_config.json", clip_sample=True)
    reload_with("unet", "config.json", upcast_attention=False, sample_size=64, dropout=0.1)   # informational fields


def test_iid_output_container_and_pipeline_contract():
    """MarigoldIIDOutput / IIDEntry semantics (marigold_iid_pipeline.py:59-161) and the channel contract
    of MarigoldIIDPipeline, without a GPU."""
    import marigold_amd as M
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import UNetConfig, TINY_VAE
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler, LCMScheduler
    props = {"target_names": ["albedo", "shading"], "albedo": {"prediction_space": "srgb"},
             "shading": {"prediction_space": "linear", "up_to_scale": True}}
    out = M.MarigoldIIDOutput(props["target_names"])
    a = torch.rand(1, 3, 5, 7)
    out.fill_entry("albedo", a, None, props)
    with pytest.raises(RuntimeError):
        out.fill_entry("albedo", a, None, props)
    with pytest.raises(KeyError):
        out.fill_entry("nope", a, None, props)
    assert not out.is_complete
    sh = torch.rand(1, 3, 5, 7) * 0.5
    out.fill_entry("shading", sh, torch.rand(1, 3, 5, 7), props)
    assert out.is_complete and out["shading"].uncertainty.shape == (3, 5, 7)
    want = (((sh[0] / sh.max()) ** (1 / 2.2)).numpy() * 255).astype(np.uint8)
    assert np.array_equal(np.asarray(out["shading"].image), np.moveaxis(want, 0, -1))
    assert [e.name for e in out] == props["target_names"] and out["albedo"].array.shape == (3, 5, 7)
    ucfg = UNetConfig(in_channels=12, out_channels=8, block_out_channels=(64, 128, 128, 128), heads=(1, 2, 2, 2),
                      cross_attention_dim=64)
    unet = UNet2DConditionModelHIP(syn.synthetic_unet_state_dict(ucfg), ucfg)
    vae = AutoencoderKLHIP(syn.synthetic_vae_state_dict(TINY_VAE), TINY_VAE)
    pipe = M.MarigoldIIDPipeline(unet, vae, DDIMScheduler(), target_properties=props, default_denoising_steps=4,
                                 default_processing_resolution=768,
                                 empty_text_embed=syn.synthetic_text_embedding(64))
    assert pipe.n_targets == 2 and pipe._target_latent_channels == 8 and pipe._pred_channels == 6
    with pytest.raises(ValueError, match="does not match"):
        M.MarigoldIIDPipeline(unet, vae, DDIMScheduler(), target_properties={"target_names": ["a"], "a": {}})
    pipe.scheduler = LCMScheduler()
    with pytest.raises(RuntimeError, match="does not support the LCMScheduler"):
        pipe._check_inference_step(4)
    # the 12 -> 64 conv_in / 64 -> 8 conv_out programs validate through the C ABI without a device
    unet.dry()
    unet.set_context(torch.zeros(1, 2, 64))
    prog = unet.denoise_program(2, 8, 16, DDIMScheduler(), 2)
    prog.seq.validate()
    assert tuple(prog.x.shape) == (2, 8, 8, 16) and tuple(prog.eps.shape) == (2, 8, 8, 16)


def test_pingpong_gemm_schedule_model():
    """Happens-before model of the ping-pong K loop that the automatic tile choice uses for N % 256 == 0 layers
    (csrc/igemm2.hip, LOOP = 2, tile variants 60-63).  The eight waves form two groups that run one barrier apart;
    every barrier is a rendezvous of all waves, so an event of one group is ordered before an event of the other
    iff a barrier lies between them.  Replays each group's program (DMA pieces, counted vmcnt waits, barriers,
    fragment reads + their lgkmcnt(0)) and checks
      RAW: a half tile is read only after BOTH groups' covering waits, with a barrier in between for the other group;
      WAR: a DMA piece goes into an LDS slot only after both groups' reads of the slot's previous tile have returned.
    Burst (variant 60) and mid-MFMA issue (variant 62) forms, several K-tile counts."""
    def program(group, KT, mid):
        ev, epoch = [], 0                       # (kind, payload, epoch)

        def add(kind, payload=None):
            ev.append((kind, payload, epoch))

        def barrier():
            nonlocal epoch
            epoch += 1

        for h in range(4):
            add("issue", (0, h, 0)); add("issue", (0, h, 1))
        add("wait", 4); barrier()
        if group == 1:
            barrier()
        w12 = 3 if mid else 4
        reads = {0: (0, 1), 1: (2,), 2: (3,), 3: ()}
        for t in range(KT):
            nxt = t + 1 < KT
            for p in range(4):
                for h in reads[p]:
                    add("read", (t, h))
                if nxt:
                    add("issue", (t + 1, p, 0))
                    if p == 3 or not mid:
                        add("issue", (t + 1, p, 1))
                if p in (0, 1):
                    add("wait", w12 if nxt else (2 if p == 0 else 0))
                if p == 3 and nxt:
                    add("wait", 4)
                barrier()
                add("lgkm")                     # s_waitcnt lgkmcnt(0): this phase's fragment reads have returned
                if nxt and mid and p < 3:
                    add("issue", (t + 1, p, 1))
                barrier()
        if group == 0:
            barrier()
        return ev, epoch

    for mid in (False, True):
        for KT in (1, 2, 3, 7):
            progs = [program(g, KT, mid) for g in (0, 1)]
            assert progs[0][1] == progs[1][1]           # both groups execute the same number of barriers
            retired_at, read_done_at, issues, reads = [{}, {}], [{}, {}], [[], []], [[], []]
            for g, (ev, _) in enumerate(progs):
                queue, pending_reads = [], []
                for idx, (kind, payload, epoch) in enumerate(ev):
                    if kind == "issue":
                        queue.append(payload)
                        issues[g].append((payload, idx, epoch))
                    elif kind == "wait":
                        for piece in queue[:max(0, len(queue) - payload)]:
                            retired_at[g].setdefault(piece, (idx, epoch))
                        del queue[:max(0, len(queue) - payload)]
                    elif kind == "read":
                        reads[g].append((payload, idx, epoch))
                        pending_reads.append(payload)
                    elif kind == "lgkm":
                        for r in pending_reads:
                            read_done_at[g][r] = (idx, epoch)
                        pending_reads = []
                assert not queue, (mid, KT, g, queue)   # everything issued is eventually waited for
            for g in (0, 1):
                for (t, h), ridx, repoch in reads[g]:
                    for o in (0, 1):
                        for piece in ((t, h, 0), (t, h, 1)):
                            widx, wepoch = retired_at[o][piece]
                            assert (widx < ridx) if o == g else (wepoch < repoch), ("RAW", mid, KT, g, o, t, h)
                for (t, h, _), iidx, iepoch in issues[g]:
                    if t < 2:
                        continue                         # first use of the slot (tile t - 2 shares stage and half)
                    for o in (0, 1):
                        didx, depoch = read_done_at[o][(t - 2, h)]
                        assert (didx < iidx) if o == g else (depoch < iepoch), ("WAR", mid, KT, g, o, t, h)



def test_rowgemm_weight_packing():
    """weights.pack_rowgemm: fragment (stage j, tile t, K step s), lane l = 32 g + mm holds w[64 j + 32 t + chan(mm)][16 s + 8 g .. + 8]
    (csrc/rowgemm.hip's operand order); the trailer holds the stage's per-channel constants; the GEGLU row order pairs
    32 value rows with their 32 gate rows."""
    import torch
    from marigold_amd import weights as Wm
    g = torch.Generator().manual_seed(5)
    n, k = 192, 320
    w, cb, lg = torch.randn(n, k, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    pk = Wm.pack_rowgemm(w, cb, lg)
    ks = k // 16
    assert pk.shape == (n // 64, (2 * ks + 1) * 1024) and pk.dtype == torch.uint8
    fr = pk[:, :2 * ks * 1024].contiguous().view(torch.bfloat16).view(n // 64, 2, ks, 2, 32, 8)
    wb = w.to(torch.bfloat16)
    chan = [(m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1) for m in range(32)]
    assert sorted(chan) == list(range(32)) and chan[:8] == [0, 1, 2, 3, 8, 9, 10, 11]
    for j in range(n // 64):
        for t in range(2):
            for mm in range(32):
                for h in range(2):
                    assert torch.equal(fr[j, t, :, h, mm, :], wb[64 * j + 32 * t + chan[mm]].view(ks, 2, 8)[:, h])
    tr = pk[:, 2 * ks * 1024:].contiguous().view(torch.float32)
    assert torch.equal(tr[:, :64].reshape(-1), cb) and torch.equal(tr[:, 64:128].reshape(-1), lg) and not tr[:, 128:].any()
    # K = 640: two slots per stage, one 32-channel tile each; the constants ride in the second slot's trailer
    k = 640
    w = torch.randn(128, k, generator=g)
    pk = Wm.pack_rowgemm(w, cb[:128], lg[:128])
    assert pk.shape == (4, 41 * 1024)
    fr = pk[:, :40 * 1024].contiguous().view(torch.bfloat16).view(2, 2, 40, 2, 32, 8)   # [j][t = slot parity][s][g][mm][i]
    wb = w.to(torch.bfloat16)
    for j in range(2):
        for t in range(2):
            for mm in range(32):
                for h in range(2):
                    assert torch.equal(fr[j, t, :, h, mm, :], wb[64 * j + 32 * t + chan[mm]].view(40, 2, 8)[:, h])
    tr = pk[:, 40 * 1024:].contiguous().view(torch.float32).view(2, 2, 256)
    assert not tr[:, 0].any() and torch.equal(tr[:, 1, :64].reshape(-1), cb[:128]) and torch.equal(tr[:, 1, 64:128].reshape(-1), lg[:128])
    order = Wm.rowgemm_geglu_order(512)
    assert sorted(order.tolist()) == list(range(512))
    assert order[:32].tolist() == list(range(32)) and order[32:64].tolist() == list(range(256, 288)) and order[64] == 32


def test_rowgemm_cross_attention_packing():
    """weights.pack_rowgemm_xattn (K = 320: scores stage + VO^T sub-stages + bias) and pack_rowgemm_xattn_ksplit (K = 640 / 1280:
    per-wave K quarters) lay the fragments out as csrc/rowgemm.hip reads them."""
    import torch
    from marigold_amd import weights as Wm
    g = torch.Generator().manual_seed(9)
    chan = [(m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1) for m in range(32)]
    # K = 320
    k = 320
    wqk, c, gg = torch.randn(64, k, generator=g), torch.randn(64, generator=g), torch.randn(64, generator=g)
    vot, bias = torch.randn(k, 64, generator=g), torch.randn(k, generator=g)
    pk = Wm.pack_rowgemm_xattn(wqk, c, gg, vot, bias)
    assert pk.numel() == (41 + 40 + 2) * 1024
    assert torch.equal(pk[:41 * 1024], Wm.pack_rowgemm(wqk, c, gg).reshape(-1))
    fr = pk[41 * 1024: 81 * 1024].view(torch.bfloat16).view(k // 64, 2, 4, 2, 32, 8)    # [jj][t][s][g][mm][i]
    vb = vot.to(torch.bfloat16)
    for jj in range(k // 64):
        for t in range(2):
            for mm in range(32):
                assert torch.equal(fr[jj, t, :, :, mm, :].reshape(-1), vb[64 * jj + 32 * t + chan[mm]])
    tr = pk[81 * 1024:].view(torch.float32)
    assert torch.equal(tr[:k], bias) and not tr[k:].any()
    # K = 1280: wave w owns channels [320 w, 320 (w + 1)) on both sides
    k = 1280
    wqk, vot, bias = torch.randn(64, k, generator=g), torch.randn(k, 64, generator=g), torch.randn(k, generator=g)
    pk = Wm.pack_rowgemm_xattn_ksplit(wqk, c, gg, vot, bias)
    kq, ksq, nt2 = k // 4, k // 64, k // 128
    n1, n2 = 4 * 2 * ksq * 1024, 4 * nt2 * 4 * 1024
    assert pk.numel() == n1 + n2 + (128 + k) * 4
    f1 = pk[:n1].view(torch.bfloat16).view(4, 2, ksq, 2, 32, 8)        # [w][t][s][g][mm][i]
    f2 = pk[n1:n1 + n2].view(torch.bfloat16).view(4, nt2, 4, 2, 32, 8)   # [w][tt][s'][g][mm][i]
    wb, vb = wqk.to(torch.bfloat16), vot.to(torch.bfloat16)
    for w in range(4):
        for t in range(2):
            for mm in (0, 5, 12, 31):
                assert torch.equal(f1[w, t, :, :, mm, :].reshape(-1), wb[32 * t + chan[mm], kq * w: kq * (w + 1)])
        for tt in (0, nt2 - 1):
            for mm in (0, 9, 31):
                assert torch.equal(f2[w, tt, :, :, mm, :].reshape(-1), vb[kq * w + 32 * tt + chan[mm]])
    fl = pk[n1 + n2:].view(torch.float32)
    assert torch.equal(fl[:64], c) and torch.equal(fl[64:128], gg) and torch.equal(fl[128:], bias)


def test_rowgemm_launch_shape_per_row_count():
    """engine.Builder.rowgemm_cfg: 12 waves while 384-row workgroups still fill the chip, then 8, then 4 with the column stages
    shared out (never for the whole-row-statistics forms); the K = 640 form always 8 waves with the columns split so that
    workgroups <= CUs."""
    from marigold_amd.engine import Builder
    T = 9216
    assert Builder.rowgemm_cfg(10 * T, 960) == dict(waves=12)
    assert Builder.rowgemm_cfg(7 * T, 960) == dict(waves=12)
    assert Builder.rowgemm_cfg(5 * T, 960) == dict(waves=8)
    c1 = Builder.rowgemm_cfg(1 * T, 960)
    assert c1["waves"] == 4 and c1["nsplit"] == 4          # 72 row blocks x 4 = 288 workgroups
    assert Builder.rowgemm_cfg(1 * T, 320, whole_rows=True) == dict(waves=4)
    assert Builder.rowgemm_cfg(2 * T, 320, xattn=True) == dict(waves=8)
    w = Builder.rowgemm_cfg(10 * 2304, 1920, K=640)
    assert w == dict(waves=8, nsplit=2)                    # 90 blocks x 2
    w = Builder.rowgemm_cfg(5 * 2304, 5120, K=640)
    assert w["waves"] == 8 and 45 * w["nsplit"] <= 256 and w["nsplit"] >= 4


def test_native_bfgs_follows_scipy():
    """csrc/bfgs.hip restates scipy 1.15's BFGS (+ DCSRCH / Wolfe-2 line searches, one evaluation per distinct point): on smooth,
    non-smooth and the alignment objective itself (a host stand-in for the device pass) the iterates agree to rounding and the
    iteration / evaluation counts are equal - except where a run ends in scipy's "precision loss" branch, whose last step
    depends on the last bit of a dot product (BLAS summation order)."""
    import ctypes
    import scipy
    import torch
    from scipy.optimize import minimize
    from marigold_amd import _lib as L, ensemble as E
    if not scipy.__version__.startswith("1.15."):
        pytest.skip(f"csrc/bfgs.hip follows scipy 1.15's _minimize_bfgs / DCSRCH; installed: {scipy.__version__}")
    lib = L.load()
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_double))

    def native(fun, x0, tol, maxiter):
        def cb(user, n_, xp, fp, gp):
            f, g = fun(np.ctypeslib.as_array(xp, (n_,)).copy())
            fp[0] = f
            for i in range(n_):
                gp[i] = g[i]
            return 0
        c = CB(cb)
        x = np.array(x0, dtype=np.float64).copy()
        fval, nit, nfev, st = ctypes.c_double(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.check(lib.mg_bfgs_minimize(ctypes.cast(c, ctypes.c_void_p), None, len(x), x.ctypes.data, tol, maxiter, ctypes.byref(fval),
                                     ctypes.byref(nit), ctypes.byref(nfev), ctypes.byref(st)), "mg_bfgs_minimize")
        return x, fval.value, nit.value, nfev.value, st.value

    def ref(fun, x0, tol, maxiter):
        cnt = [0]

        def f(x):
            cnt[0] += 1
            return fun(x)
        r = minimize(f, x0, jac=True, method="BFGS", tol=tol, options={"maxiter": maxiter})
        return r.x, float(r.fun), r.nit, cnt[0], r.status

    def rosen(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    a, b = native(rosen, np.array([-1.2, 1, 0.5, -0.3, 2.0]), 1e-6, 50), ref(rosen, np.array([-1.2, 1, 0.5, -0.3, 2.0]), 1e-6, 50)
    assert a[2:] == b[2:] and np.allclose(a[0], b[0], rtol=1e-9, atol=1e-12) and abs(a[1] - b[1]) <= 1e-9 * abs(b[1])
    kink = lambda x: (np.abs(x).sum() + 0.5 * (x ** 2).sum(), np.sign(x) + x)      # noqa: E731 - non-smooth: the Wolfe-2 fall-back, precision loss
    a, b = native(kink, np.array([0.3, -0.7, 1.1]), 1e-6, 50), ref(kink, np.array([0.3, -0.7, 1.1]), 1e-6, 50)
    assert a[2:] == b[2:] and np.allclose(a[0], b[0], atol=1e-12)

    class FakeBackend:
        def __init__(self, n, seed):
            self.n, self.rng = n, np.random.default_rng(seed)

        def stats(self):
            n, r = self.n, self.rng
            A = r.normal(size=(n, n))
            self.dmn, self.dmx = r.uniform(0.0, 0.3, n), r.uniform(0.7, 1.0, n)
            return r.uniform(0, 0.05, n), r.uniform(0.9, 1.1, n), r.normal(size=n) * 0.05 + 0.5, A @ A.T / n * 0.01 + np.eye(n) * 0.002

        def regulariser(self, s32, t32):
            a, b = self.dmn.astype(np.float32) * s32 + t32, self.dmx.astype(np.float32) * s32 + t32
            k = (self.n - 1) // 2
            return (float(np.sort(a)[k]), float(np.sort(b)[k]), self.dmn.astype(np.float32).astype(np.float64),
                    self.dmx.astype(np.float32).astype(np.float64))

    same = total = 0
    for seed in range(6):
        for n in (10, 4, 20):
            al = E.DepthAligner(torch.zeros(n, 1, 4, 4), True, True, "median", 0.02, backend=FakeBackend(n, seed))
            p0 = al.init_param()
            a, b = native(al.reference_fd_objective, p0, 1e-6, 50), ref(al.reference_fd_objective, p0, 1e-6, 50)
            assert np.allclose(a[0], b[0], rtol=1e-7, atol=1e-10) and a[2] == b[2] and abs(a[3] - b[3]) <= 2, (seed, n, a[1:], b[1:])
            total += 1
            same += a[2:] == b[2:]
    assert same >= total - 2, (same, total)


@pytest.mark.parametrize("B,heads,Ntok", [(10, 5, 9216), (10, 10, 2304), (1, 5, 9216), (13, 5, 1024), (3, 23, 1280), (2, 5, 9216), (7, 3, 512),
                                          (1, 1, 256), (16, 16, 4096)])
@pytest.mark.parametrize("split", [0, 1, 2])
def test_flash_attention_key_split_plan(B, heads, Ntok, split):
    """MG_OP_FLASH_ATTN64's hand-placed kernel splits the blocks of 256 queries left over beyond a multiple of the CU count along
    the keys (csrc/flash4w.hip).  Host-side properties of that plan (mg_flash4w_plan_test, no device): every block is either whole
    or split, the workgroups' ranges tile the split blocks' key tiles exactly once, every piece of a block has at least the four
    tiles the stream needs, a block has at most four pieces (the workspace holds four partial results per block)."""
    import ctypes
    from marigold_amd import _lib as L, ops as O
    lib = L.load()
    n_cu, nkt = 256, Ntok // 64
    nb = (Ntok // 256) * heads * B
    out = (ctypes.c_int * 3)()
    bounds = (ctypes.c_uint * 1026)()
    assert lib.mg_flash4w_plan_test(B, heads, Ntok, n_cu, O.FLASH_WS_BYTES, split, out, bounds) == 0
    n_full, n_rem, nwg = out[0], out[1], out[2]
    assert n_full + n_rem == nb and n_full % n_cu == 0 or n_rem == 0
    if split == 2 or nb % n_cu == 0:
        assert n_rem == 0 and nwg == 0
    if split == 1 and nb % n_cu != 0 and nkt >= 8:
        assert n_rem == nb % n_cu and nwg > n_rem
    if n_rem == 0:
        return
    assert nwg <= 2 * n_cu
    b = [bounds[r] for r in range(nwg + 1)]
    assert b[0] == 0 and b[-1] == n_rem * nkt and all(x <= y for x, y in zip(b, b[1:]))
    pieces = [0] * n_rem
    for lo, hi in zip(b, b[1:]):
        s = lo
        while s < hi:                         # the kernel's walk over its range, block by block
            blk, t0 = divmod(s, nkt)
            t1 = min(nkt, t0 + hi - s)
            assert t1 - t0 >= 4, (lo, hi, blk, t0, t1)
            pieces[blk] += 1
            s += t1 - t0
    assert min(pieces) >= 1 and max(pieces) <= 4, (min(pieces), max(pieces))
    # without a workspace nothing is split
    assert lib.mg_flash4w_plan_test(B, heads, Ntok, n_cu, 0, split, out, None) == 0 and out[1] == 0 and out[0] == nb


def test_generated_instruction_streams_are_in_sync(tmp_path):
    """The hand-placed instruction streams are written by generators (csrc/gen_k4w.py, gen_cp4w.py, gen_fa4w.py) and committed as
    .inc files (the build regenerates them when a generator is newer): the committed files must be what the generators write."""
    import subprocess
    import sys
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marigold_amd", "csrc")
    for gen, inc, to_stdout in (("gen_k4w.py", "igemm2_k4w.inc", True), ("gen_cp4w.py", "conv_patch4w.inc", True), ("gen_fa4w.py", "flash4w.inc", False)):
        out = tmp_path / inc
        if to_stdout:
            r = subprocess.run([sys.executable, os.path.join(csrc, gen)], capture_output=True, text=True, cwd=csrc, timeout=300)
            assert r.returncode == 0, r.stderr[-500:]
            text = r.stdout
        else:
            r = subprocess.run([sys.executable, os.path.join(csrc, gen), str(out)], capture_output=True, text=True, cwd=csrc, timeout=300)
            assert r.returncode == 0, r.stderr[-500:]
            text = out.read_text()
        assert text == open(os.path.join(csrc, inc)).read(), f"{inc} is not what {gen} writes"


def test_conv3x3_groupnorm_byproduct_slots():
    """mg_conv3x3_gn_slots (host-only): which MG_OP_CONV3X3 launches can leave the GroupNorm partial sums of their output, and
    how many table slots per image they fill (one per tile, four per tile in the sub-pixel form); 0 for the tiles that cannot."""
    from marigold_amd import ops
    def slots(**kw):
        base = dict(B=10, H=384, W=384, C0=256, N=256)
        base.update(kw)
        return ops.conv3x3_gn_slots(ops.conv3x3(0, 0, 0, **base))
    assert slots(variant=9) == 32 * 24                          # 12 x 16 pixel tiles
    assert slots(variant=9, subpix=True, wz=1) == 4 * 32 * 24
    assert slots(variant=8, N=128, H=768, W=768) == 32 * 48    # 24 x 16
    assert slots(variant=1, subpix=True, wz=1) == 4 * 24 * 24  # 16 x 16
    assert slots(variant=8, N=128, H=50, W=37) == 3 * 3        # ragged
    assert slots(variant=6, N=320, C0=320) == 0 and slots(variant=10) == 0 and slots(variant=11, N=320, C0=320) == 0
    assert slots(variant=9, N=128) == 0                         # the 256-channel tile needs N % 256 == 0
