#!/usr/bin/env python
"""Headline benchmark: depth maps/sec @768x768, ensemble_size=10, 10 DDIM steps (BASELINE.json).

One "step" = one full ``MarigoldDepthPipeline.__call__`` on one synthetic 768x768 image: VAE encode
(once), E=10 members x T=10 (UNet forward + DDIM update) as ONE native program, VAE decode of all
members, on-device ensembling, D2H of the final [768,768] fp32 map (color_map=None, like
script/depth/infer.py).  The image is already resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the E members of every map are sharded over the ranks, collected with
ONE gather (RCCL over xGMI) and aggregated on rank 0 ("strong" scaling: the work per map is fixed).
MARIGOLD_BENCH_FORCE_DIST=1 takes the SAME code path with one rank (``init_process_group("nccl", device_id=...)``,
``enable_member_parallel``, the rooted gather to self, the all-reduce of the timing): the RCCL path executed on one GPU
(tests/test_gpu_pipeline.py::test_bench_nccl_path_single_rank); the line then carries a ``collective`` object.
Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).  Weights are seeded synthetic
tensors in the real SD-v2 architecture (no checkpoints / network here); arithmetic is bf16 with fp32
accumulation.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


SURVEY_TFLOP_PER_MAP = 273.92   # E = 10, T = 10, 768 x 768, reference-faithful FLOP count


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="maps on the GPU at a time (pipeline.map_images: independent maps on concurrent HIP streams); "
                         "0 = the pipeline's default (2)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ensemble", type=int, default=10)
    ap.add_argument("--denoise", type=int, default=10)
    ap.add_argument("--res", type=int, default=768)
    ap.add_argument("--kind", default="depth", choices=["depth", "normals", "iid"],
                    help="iid = the appearance model (albedo + material: UNet 12 -> 8 latent channels)")
    ap.add_argument("--scheduler", default="ddim", choices=["ddim", "lcm"], help="ddim = v1-1 (trailing, zero-SNR); lcm = depth-lcm-v1-0")
    ap.add_argument("--tiny", action="store_true", help="tiny architecture (plumbing check only)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit operand type of the engine (fp32 accumulation): bf16 = the headline build; fp16 = libmarigold_hip_f16.so, "
                         "the reference's --fp16 arithmetic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the denoising loop as a hipGraph")
    ap.add_argument("--dump-ops", default="", help="write the per-op timing table to this file")
    ap.add_argument("--cpu-baseline-only", default="", help=argparse.SUPPRESS)  # internal: child mode
    ap.add_argument("--cpu-baseline-timeout", type=int, default=200)
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel class from the rocprofv3 PMC passes of THIS command
    (scripts/gpu_round.sh pmc: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as
    /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950's wide coalesced reads),
    committed as profiles/r<N>_pmc_hbm_traffic.json (the newest round present is used).  PMC counters cannot be read
    from inside the timed process, so the figure is the one measured offline for the same workload; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None
    path = files[-1]
    try:
        with open(path) as f:
            j = json.load(f)
        t = j.get(kernel_class)
        rel = os.path.relpath(path, ROOT)
        return None if t is None else {"bytes_per_launch": t["bytes_per_launch"], "launches": t["launches"],
                                       "source": f"{rel} (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE)", "build": j.get("_build") or {}}
    except (OSError, ValueError, KeyError):
        return None


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


class GpuTelemetry:
    """Clock / power state of the GPU this rank runs on, sampled from sysfs (amdgpu: pp_dpm_sclk / pp_dpm_mclk mark the
    active level with '*', hwmon power1_average in microwatts, freq1_input in Hz) by a host thread while the timed region
    runs.  Box-speed evidence for the bench line (same binary, 353-402 ms per map depending on the box);
    everything is best effort - a missing file just leaves its field out."""

    def __init__(self, index=0):
        import glob
        self.dev = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if _read(os.path.join(c, "pp_dpm_sclk")) is not None]
        if cards:
            self.dev = cards[min(index, len(cards) - 1)]
        self.samples = []
        self._stop = None

    @staticmethod
    def _active_mhz(text):
        if not text:
            return None
        for ln in text.splitlines():
            if ln.strip().endswith("*"):
                try:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except (IndexError, ValueError):
                    return None
        return None

    def sample(self):
        if self.dev is None:
            return None
        import glob
        d = {"sclk_mhz": self._active_mhz(_read(os.path.join(self.dev, "pp_dpm_sclk"))),
             "mclk_mhz": self._active_mhz(_read(os.path.join(self.dev, "pp_dpm_mclk")))}
        for hw in glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")):
            for key, fn, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6),
                                   ("gfx_mhz", "freq1_input", 1e-6), ("temp_c", "temp1_input", 1e-3),
                                   ("power_cap_w", "power1_cap", 1e-6)):
                v = _read(os.path.join(hw, fn))
                if v is not None and d.get(key) is None:
                    try:
                        d[key] = float(v) * scale
                    except ValueError:
                        pass
        return d

    def start(self, period=0.02):
        import threading
        if self.dev is None:
            return
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                s = self.sample()
                if s:
                    self.samples.append(s)
                self._stop.wait(period)
        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()

    def stop(self):
        if self._stop is None:
            return None
        self._stop.set()
        self._thr.join(timeout=1.0)
        out = {"samples": len(self.samples)}
        for k in ("sclk_mhz", "gfx_mhz", "mclk_mhz", "power_w", "temp_c", "power_cap_w"):
            vals = [s[k] for s in self.samples if s.get(k) is not None]
            if vals:
                out[k] = {"mean": round(sum(vals) / len(vals), 1), "min": round(min(vals), 1), "max": round(max(vals), 1)}
        return out


def calibration(dev):
    """Box-speed index measured in this process before the warm-up: a fixed 4096^3 bf16 GEMM on this
    library's own implicit-GEMM kernel (and on torch.matmul = hipBLASLt, for a second opinion), a 1 GiB device-to-device
    copy, the idle clock / power state.  Random operands (zero-filled ones clock ~20 % higher)."""
    import torch
    from marigold_amd import ops as O
    out = {}

    def timeit(fn, iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    try:
        n = 4096
        g = torch.Generator(device="cpu").manual_seed(7)
        a = (torch.rand(n, n, generator=g) * 2 - 1).to(dev, torch.bfloat16)
        w = (torch.rand(n, n, generator=g) * 2 - 1).to(dev, torch.bfloat16)
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        op = O.linear(a, w, c, M=n, K=n, N=n)
        ms = min(timeit(lambda: O.launch(op), 10) for _ in range(3))
        out["gemm4096_bf16_tflops"] = round(2.0 * n ** 3 / ms / 1e9, 1)
        ms = min(timeit(lambda: torch.matmul(a, w.t(), out=c), 10) for _ in range(3))
        out["gemm4096_bf16_tflops_hipblaslt"] = round(2.0 * n ** 3 / ms / 1e9, 1)
        del a, w, c
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
        dst = torch.empty_like(src)
        ms = min(timeit(lambda: dst.copy_(src), 5) for _ in range(3))
        out["copy_1gib_gbs"] = round(2.0 * (1 << 30) / ms / 1e6, 1)   # bytes read + written
        del src, dst
        # the shader clock this box sustains under matrix-core load, measured IN the kernel (s_memtime against the 100 MHz
        # s_memrealtime around a fixed MFMA chain, one wave per SIMD on every CU; csrc/runtime.hip::mg_clock_probe) - the
        # sysfs sensors do not answer on every box of the pool (round 3: constant idle values through the run)
        import ctypes
        from marigold_amd import _lib as L
        mhz, tf = ctypes.c_double(0.0), ctypes.c_double(0.0)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for key, zero in (("shader_mhz_under_mfma_load", 0), ("shader_mhz_under_mfma_load_zero_operands", 1)):
            if L.load().mg_clock_probe(stream, zero, ctypes.byref(mhz), ctypes.byref(tf)) == 0:
                out[key] = round(mhz.value, 1)
                out[key.replace("shader_mhz", "mfma_chain_tflops")] = round(tf.value, 1)
        torch.cuda.empty_cache()
        out["device"] = torch.cuda.get_device_name(dev)
        out["idle"] = GpuTelemetry(dev.index or 0).sample()
    except Exception as e:  # noqa: BLE001 - reporting only
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def cpu_baseline(args, members_path):
    """The CPU oracle (restatement of the reference's diffusers path) timed on the host cores, on a
    bounded sample: 1 UNet forward at the full 96x96 latent, VAE encode/decode at 256x256 scaled by
    pixel count, and the reference's ensemble_depth algorithm (BFGS evaluation count from a 192x192
    run x the cost of one evaluation at full size).  Reported, never used by the product path."""
    import numpy as np
    import torch
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE, UNetConfig, VAEConfig
    from marigold_amd.util.host import cpu_model, usable_cores
    from oracle import ensemble as oens
    from oracle.sd2_unet import UNet2DConditionModel
    from oracle.sd2_vae import AutoencoderKL
    ucfg, vcfg = (TINY_UNET, TINY_VAE) if args.tiny else (UNetConfig(), VAEConfig())
    n_dec = 1
    if args.kind == "iid":
        import dataclasses
        ucfg, n_dec = dataclasses.replace(ucfg, in_channels=12, out_channels=8), 2   # one VAE decode per modality
    usd, vsd = syn.synthetic_unet_state_dict(ucfg), syn.synthetic_vae_state_dict(vcfg)
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    members = torch.from_numpy(np.load(members_path)) if members_path and os.path.exists(members_path) else None
    cores = min(usable_cores(), 64)   # one socket's worth: torch's CPU kernels stop scaling beyond
    torch.set_num_threads(cores)
    E, T, res = args.ensemble, args.denoise, args.res
    lat = res // 8
    t = {}
    with torch.no_grad():
        unet = UNet2DConditionModel(in_channels=ucfg.in_channels, out_channels=ucfg.out_channels,
                                    block_out_channels=ucfg.block_out_channels, attention_head_dim=ucfg.heads,
                                    cross_attention_dim=ucfg.cross_attention_dim).eval()
        unet.load_state_dict(usd)
        x = torch.randn(1, ucfg.in_channels, lat, lat)
        t0 = time.perf_counter()
        unet(x, torch.tensor(999), ctx)
        t["unet_fwd"] = time.perf_counter() - t0
        del unet
        vae = AutoencoderKL(block_out_channels=vcfg.block_out_channels).eval()
        vae.load_state_dict(vsd)
        sres = min(res, 256)
        scale = (res / sres) ** 2
        img = torch.rand(1, 3, sres, sres) * 2 - 1
        t0 = time.perf_counter()
        h = vae.quant_conv(vae.encoder(img))
        t["vae_encode"] = (time.perf_counter() - t0) * scale
        z = h[:, :4]
        t0 = time.perf_counter()
        vae.decoder(vae.post_quant_conv(z))
        t["vae_decode"] = (time.perf_counter() - t0) * scale
        del vae
        t["ensemble"] = 0.0
        n_eval = 0
        if E > 1 and members is not None and args.kind == "depth":
            m = members.float().cpu()
            small = torch.nn.functional.interpolate(m, (192, 192), mode="nearest-exact")
            calls = [0]
            orig = oens.depth_cost

            def counting(*a, **k):
                calls[0] += 1
                return orig(*a, **k)
            oens.depth_cost = counting
            try:
                _, _, p = oens.ensemble_depth(small, True, True, return_param=True)
            finally:
                oens.depth_cost = orig
            n_eval = calls[0]
            t0 = time.perf_counter()
            for _ in range(5):
                orig(p, m, True, True, "median", 0.02)
            t["ensemble"] = (time.perf_counter() - t0) / 5 * n_eval
    per_map = E * (t["vae_encode"] + T * t["unet_fwd"] + n_dec * t["vae_decode"]) + t["ensemble"]
    return {"value": 1.0 / per_map, "unit": f"{args.kind} maps/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": (f"CPU oracle fp32: 1 UNet fwd @{lat}x{lat} latent ({t['unet_fwd']:.2f}s) + VAE enc/dec "
                       f"@{min(res, 256)}^2 scaled x{scale:.0f} by pixels ({t['vae_encode']:.1f}s/{t['vae_decode']:.1f}s) "
                       f"+ reference ensemble_depth ({n_eval} BFGS cost evals from a 192^2 run x one eval @{res}^2 = "
                       f"{t['ensemble']:.1f}s); extrapolated as E*(enc+T*unet+dec)+ens = {per_map:.0f}s/map"),
            "seconds_per_map": per_map}


def main():
    args = parse()
    if args.cpu_baseline_only:
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args, args.cpu_baseline_only)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run` with one rank
        # per GPU (never fall back to one GPU: a run that cannot get N ranks fails)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU"
    n_dev = torch.cuda.device_count()
    shared_gpu = os.environ.get("MARIGOLD_BENCH_SHARE_GPU") == "1"   # test rigs only: several ranks on one device
    assert shared_gpu or world <= n_dev, f"--gpus {args.gpus} but only {n_dev} GPU(s) visible"
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    force_dist = os.environ.get("MARIGOLD_BENCH_FORCE_DIST") == "1"   # one rank through the multi-rank code path (RCCL on one GPU)
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "WORLD_SIZE" not in os.environ:   # forced, without a launcher: rendezvous with ourselves
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("gloo" if shared_gpu else "nccl", **({} if shared_gpu else {"device_id": dev}))
        seen = torch.ones(1, device="cpu" if shared_gpu else dev)
        dist.all_reduce(seen)
        assert int(seen.item()) == args.gpus == dist.get_world_size(), \
            f"the process group sees {int(seen.item())} ranks, --gpus {args.gpus}"
        world = dist.get_world_size()

    from marigold_amd.util.host import usable_cores
    torch.set_num_threads(max(1, min(32, usable_cores() // max(1, world))))   # weight synthesis is the only host-heavy part
    import marigold_amd as M
    from marigold_amd import opstats, synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE, UNetConfig, VAEConfig
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler, LCMScheduler

    ucfg, vcfg = (TINY_UNET, TINY_VAE) if args.tiny else (UNetConfig(), VAEConfig())
    if args.kind == "iid":
        import dataclasses
        ucfg = dataclasses.replace(ucfg, in_channels=12, out_channels=8)
    t0 = time.perf_counter()
    usd = syn.synthetic_unet_state_dict(ucfg)
    vsd = syn.synthetic_vae_state_dict(vcfg)
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    cls = {"depth": M.MarigoldDepthPipeline, "normals": M.MarigoldNormalsPipeline, "iid": M.MarigoldIIDPipeline}[args.kind]
    extra = {}
    if args.kind == "iid":
        extra["target_properties"] = {"target_names": ["albedo", "material"], "albedo": {"prediction_space": "srgb"},
                                      "material": {"prediction_space": "stack"}}
    cdt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    pipe = cls(unet=UNet2DConditionModelHIP(usd, ucfg, compute_dtype=cdt), vae=AutoencoderKLHIP(vsd, vcfg, compute_dtype=cdt),
               scheduler=DDIMScheduler() if args.scheduler == "ddim" else LCMScheduler(), empty_text_embed=ctx,
               default_denoising_steps=args.denoise,
               default_processing_resolution=0, **extra).to(dev)
    if dist_on:
        pipe.enable_member_parallel(root=0, force_collective=force_dist)
    if rank == 0:
        log(f"[bench] synthetic weights + pipeline ready in {time.perf_counter() - t0:.1f}s "
            f"(host cores {os.cpu_count()})")
    calib = calibration(dev) if rank == 0 else None
    if rank == 0:
        log(f"[bench] calibration: {calib}")
    img = syn.synthetic_image(args.res, args.res, seed=0).to(dev)   # resident in HBM before timing
    torch.manual_seed(2024)
    kw = dict(denoising_steps=args.denoise, ensemble_size=args.ensemble, processing_res=0,
              match_input_res=True, show_progress_bar=False)
    if args.kind == "depth":
        kw["color_map"] = None

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # maps in flight: K steps = K maps handed to pipe.map_images, which keeps up to `in_flight` of them on the GPU (each on its own
    # engine replica and HIP stream; every map is the same computation as a lone pipe(img) call, bit for bit)
    # (with several ranks every rank runs the same lanes; the gathers are issued in map order on all of them - pipeline._Turnstile)
    in_flight = args.in_flight if args.in_flight > 0 else pipe.maps_in_flight_for(args.ensemble)

    def run_maps(k):
        last = None
        for last in pipe.map_images([img] * k, in_flight=in_flight, **kw):
            pass
        return last

    out = None
    for i in range(args.warmup):
        t1 = time.perf_counter()
        if i == 0 and in_flight > 1:
            out = pipe(img, **kw)   # the first map alone: programs built, kernels loaded and their attributes set by ONE thread
        out = run_maps(in_flight)   # one map per lane: every replica's programs built
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] warmup {i}: {time.perf_counter() - t1:.3f}s ({in_flight} map(s))")
    if args.graph:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for prog in pipe.unet._programs.values():
                prog.seq.capture()
        torch.cuda.synchronize()
        out = pipe(img, **kw)
    barrier()
    t1 = time.perf_counter()
    out = run_maps(args.steps)
    barrier()
    dt = time.perf_counter() - t1
    latency_ms = None
    if in_flight > 1:   # the same maps one at a time: what a single request waits (not the headline)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(2):
            pipe(img, **kw)
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - t2) / 2 * 1e3
    tt = torch.tensor([dt], device="cpu" if (dist_on and dist.get_backend() == "gloo") else dev, dtype=torch.float64)
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if rank == 0:
        res = {"depth": lambda: out.depth_np, "normals": lambda: out.normals_np,
               "iid": lambda: out["albedo"].array}[args.kind]()
        assert res is not None and res.shape[-2:] == (args.res, args.res)
        import numpy as np
        assert np.isfinite(res).all()

    # ---- per-kernel-class roofline from HIP-event timings of every launch (rank 0) --------------
    kernels, roof, stages, flops_per_map = {}, None, {}, 0
    if rank == 0 and not args.no_profile:
        try:
            progs = [("denoise", p.seq) for p in pipe.unet._programs.values()] + \
                    [(f"vae.{k[0]}", v[0]) for k, v in pipe.vae._programs.items()]
            all_ops, all_ms, rows = [], [], []
            for name, seq in progs:
                cap = seq._captured
                if cap:   # profile the plain launch sequence, not the graph
                    continue
                ms = seq.profile()
                all_ops += seq.ops
                all_ms += ms
                stages[name] = {"ms": round(sum(ms), 3), "gflop": round(opstats.program_flops(seq.ops) / 1e9, 1),
                                "launches": len(ms)}
                rows += [(name, lab, m, op) for lab, m, op in zip(seq.labels, ms, seq.ops)]
            kernels = opstats.summarize(all_ops, all_ms)
            flops_per_map = sum(d["flops"] for d in kernels.values())
            if args.dump_ops:
                os.makedirs(os.path.dirname(os.path.abspath(args.dump_ops)), exist_ok=True)
                with open(args.dump_ops, "w") as f:
                    f.write("stage\tlabel\tclass\tms\tGFLOP\tMB\tTFLOP/s\tGB/s\n")
                    for name, lab, m, op in rows:
                        c, fl, by = opstats.op_cost(op)
                        s = max(m, 1e-6) * 1e-3
                        f.write(f"{name}\t{lab}\t{c}\t{m:.4f}\t{fl / 1e9:.2f}\t{by / 1e6:.2f}\t{fl / s / 1e12:.1f}\t{by / s / 1e9:.0f}\n")
            if kernels:
                dom = max(kernels, key=lambda c: kernels[c]["ms"])
                d = kernels[dom]
                if opstats.BOUND.get(dom) == "mfma":
                    roof = {"bound": "mfma", "kernel": dom, "achieved": round(d["tflops"], 2),
                            "peak": opstats.MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(d["tflops"] / opstats.MFMA_PEAK_TFLOPS, 4), "traffic": None,
                            "launches_per_map": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                            "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 2)}
                else:
                    roof = {"bound": "hbm", "kernel": dom, "achieved": round(d["gbs"], 1),
                            "peak": opstats.HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(d["gbs"] / opstats.HBM_PEAK_GBS, 4), "traffic": None,
                            "launches_per_map": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4)}
                # the PMC passes were collected for the headline workload only
                try:
                    headline = (args.kind, args.ensemble, args.denoise, args.res, args.scheduler, args.tiny, args.dtype) == \
                        ("depth", 10, 10, 768, "ddim", False, "bf16")
                    t = pmc_traffic(dom) if headline else None
                    if t is not None:
                        roof["traffic"] = round(t["bytes_per_launch"])      # HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE)
                        roof["traffic_unit"] = "bytes/launch"
                        roof["traffic_launches_profiled"] = t["launches"]
                        roof["traffic_source"] = t["source"]
                        roof["algorithmic_bytes_per_launch"] = round(kernels[dom]["bytes"] / kernels[dom]["launches"])
                        # the counters are collected offline (separate rocprofv3 --pmc passes): say which build they belong to and
                        # whether that is the build running now (content hash of the kernel sources + program builder + tuning table)
                        from marigold_amd.util.host import build_fingerprint
                        now = build_fingerprint()
                        roof["traffic_build"] = t["build"].get("fingerprint")
                        roof["traffic_git_head"] = t["build"].get("git_head")
                        roof["build_fingerprint"] = now
                        roof["traffic_stale"] = t["build"].get("fingerprint") != now
                except Exception as e:  # noqa: BLE001 - a reporting nicety must never cost the benchmark line
                    log(f"[bench] traffic annotation skipped: {e}")
                for d in kernels.values():
                    for k in ("ms", "tflops", "gbs"):
                        d[k] = round(d[k], 3)
        except Exception as e:  # noqa: BLE001 - the per-kernel breakdown is reporting; the benchmark line must survive it
            log(f"[bench] per-kernel profile failed: {type(e).__name__}: {e}")
            kernels, roof, stages, flops_per_map = {}, None, {}, 0

    # ---- the parts of a map that are not native programs: test-time ensembling (host BFGS driving two fused kernels
    # per cost evaluation) and, with several ranks, the single gather of the members -------------------------------
    if not args.no_profile and args.ensemble > 1:
        try:
            import marigold_amd.dist as mdist
            from marigold_amd import ensemble as ens
            C = {"depth": 1, "normals": 3, "iid": 6}[args.kind]
            local_n = len(mdist.shard_members(args.ensemble, world, rank)) if world > 1 else args.ensemble
            if dist_on:
                buf = torch.rand(local_n, C, args.res, args.res, device=dev)
                barrier()
                t1 = time.perf_counter()
                for _ in range(3):
                    got = mdist.gather_members(buf, args.ensemble, (C, args.res, args.res), dev, None, 0, force=force_dist)
                barrier()
                if rank == 0 and world == 1:   # the forced single-rank gather returns the members it was given
                    assert got is not buf and torch.equal(got, buf), "forced single-rank gather changed the members"
                if rank == 0:
                    stages["gather"] = {"ms": round((time.perf_counter() - t1) / 3 * 1e3, 3),
                                        "bytes_to_root": (args.ensemble - local_n) * C * args.res * args.res * 4}
            if rank == 0 and args.kind == "depth":
                rgbn = (img.float() / 255.0 * 2.0 - 1.0).expand(max(1, local_n), -1, -1, -1)   # a batch size whose programs exist
                members = pipe.single_infer(rgbn, args.denoise, None)
                if members.shape[0] < args.ensemble:
                    members = members.repeat(-(-args.ensemble // members.shape[0]), 1, 1, 1)[:args.ensemble]
                _, _, info = ens.ensemble_depth(members, True, True, return_info=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    ens.ensemble_depth(members, True, True)
                torch.cuda.synchronize()
                stages["ensemble"] = {"ms": round((time.perf_counter() - t1) / 3 * 1e3, 3), "cost_evaluations": int(info["n_eval"]),
                                      "bfgs_iterations": int(info["n_iter"])}
        except Exception as e:  # noqa: BLE001 - reporting only
            log(f"[bench] ensemble / gather timing skipped: {type(e).__name__}: {e}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle runs in a child process under a hard timeout: it is a reported baseline and must
        # never be able to stall the benchmark (or share threads / memory with the timed process)
        import subprocess
        import tempfile
        mpath = os.path.join(tempfile.gettempdir(), f"marigold_bench_members_{os.getpid()}.npy")
        if args.ensemble > 1 and args.kind == "depth":
            rgb = (img.float() / 255.0 * 2.0 - 1.0).cpu().expand(args.ensemble, -1, -1, -1)
            import numpy as np
            np.save(mpath, pipe.single_infer(rgb, args.denoise, None).float().cpu().numpy())
        t1 = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", mpath, "--ensemble", str(args.ensemble),
               "--denoise", str(args.denoise), "--res", str(args.res), "--kind", args.kind] + (["--tiny"] if args.tiny else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_baseline_timeout,
                               env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            got = [ln for ln in r.stdout.splitlines() if ln.startswith("CPU_BASELINE ")]
            cpu = json.loads(got[-1][len("CPU_BASELINE "):]) if got else {"value": None, "error": r.stderr[-400:]}
        except subprocess.TimeoutExpired:
            cpu = {"value": None, "error": f"cpu baseline exceeded {args.cpu_baseline_timeout}s and was stopped"}
        finally:
            if os.path.exists(mpath):
                os.remove(mpath)
        log(f"[bench] cpu baseline took {time.perf_counter() - t1:.1f}s")

    if rank == 0:
        value = args.steps / dt
        headline_cfg = (args.kind, args.ensemble, args.denoise, args.res, args.scheduler, args.tiny) == ("depth", 10, 10, 768, "ddim", False)
        if roof is not None and roof.get("bound") == "mfma" and headline_cfg and world == 1:
            roof["frac_survey"] = round(SURVEY_TFLOP_PER_MAP * value / opstats.MFMA_PEAK_TFLOPS, 4)   # whole map, SURVEY 8(d)'s FLOP count
        line = {
            "metric": f"{args.kind} maps/sec @{args.res}x{args.res}, ens={args.ensemble}, {args.denoise} {args.scheduler.upper()} steps",
            "value": round(value, 4), "unit": f"{args.kind} maps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"marigold-{args.kind}-v1-1 architecture (SD-v2 UNet 865.9M + AutoencoderKL), "
                                    f"{args.res}x{args.res}, ensemble_size={args.ensemble}, {args.denoise} "
                                    + ("DDIM steps (trailing, zero-SNR, v-prediction)" if args.scheduler == "ddim" else "LCM steps")
                                    + ", seeded synthetic weights"
                                    + (" [TINY ARCH - plumbing only]" if args.tiny else "")),
                       "members_per_gpu": -(-args.ensemble // world), "parallelism": f"member-parallel x{world}",
                       "hipgraph": bool(args.graph), "maps_in_flight": in_flight},
            # one map at a time on one stream (None when that is what the headline ran); the per-kernel tables below
            # ("stages", "kernels", "roofline") are timed that way too - launch by launch with HIP events, one map in flight
            "latency_ms_per_map": None if latency_ms is None else round(latency_ms, 2),
            "roofline": roof,
            "collective": ({"backend": dist.get_backend(), "world_size": world, "forced_single_rank": bool(force_dist and world == 1),
                            "gathers_per_map": 1} if dist_on else None),
            "cpu_baseline": cpu,
            "calibration": calib,
            # whole-map figures only where rank 0 ran the whole map (its programs cover its own members only)
            "algorithmic_tflop_per_map": round(flops_per_map / 1e12, 2) if world == 1 else None,
            "pipeline_tflops": round(flops_per_map * value / 1e12, 1) if world == 1 else None,
            "pipeline_mfma_frac": round(flops_per_map * value / 1e12 / opstats.MFMA_PEAK_TFLOPS, 4) if world == 1 else None,
            # the same against SURVEY.md section 8(d)'s reference-faithful count (273.92 TFLOP per map at the headline configuration):
            # comparable across rounds however much work the engine folds away (sub-pixel up-sampling, collapsed cross-attention)
            "pipeline_mfma_frac_survey": (round(SURVEY_TFLOP_PER_MAP * value / opstats.MFMA_PEAK_TFLOPS, 4)
                                          if world == 1 and headline_cfg else None),
            "stages": stages,
            "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        barrier()   # rank 0 spends a few seconds more (per-op profile): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
