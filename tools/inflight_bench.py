"""Maps in flight: N complete pipelines (own programs, workspaces, split-K workspaces) on N HIP streams, one host thread each.

One map's kernels leave the chip partly idle wherever a launch is a single lockstep round of workgroups (K loop, then an
HBM-bound epilogue), has a part-filled last round, or is a host-driven chain (the ensembling optimiser); a second,
independent map on another stream fills those holes.  This tool measures what that is worth on one box: K maps one after
the other on one stream against the same K maps shared out over N streams, interleaved rounds, and checks that every
concurrent map equals the sequential one bit for bit (same generator seed per map).

    python tools/inflight_bench.py --in-flight 2 --maps 6 --rounds 2
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--maps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--ensemble", type=int, default=10)
    ap.add_argument("--denoise", type=int, default=10)
    ap.add_argument("--res", type=int, default=768)
    ap.add_argument("--tiny", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import marigold_amd as M
    from marigold_amd import synthetic as syn
    from marigold_amd.arch import TINY_UNET, TINY_VAE, UNetConfig, VAEConfig
    from marigold_amd.modules import AutoencoderKLHIP, UNet2DConditionModelHIP
    from marigold_amd.schedulers import DDIMScheduler

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ucfg, vcfg = (TINY_UNET, TINY_VAE) if args.tiny else (UNetConfig(), VAEConfig())
    usd, vsd = syn.synthetic_unet_state_dict(ucfg), syn.synthetic_vae_state_dict(vcfg)
    ctx = syn.synthetic_text_embedding(ucfg.cross_attention_dim)
    N = args.in_flight
    pipes = [M.MarigoldDepthPipeline(unet=UNet2DConditionModelHIP(usd, ucfg), vae=AutoencoderKLHIP(vsd, vcfg),
                                     scheduler=DDIMScheduler(), empty_text_embed=ctx, default_denoising_steps=args.denoise,
                                     default_processing_resolution=0).to(dev) for _ in range(N)]
    img = syn.synthetic_image(args.res, args.res, seed=0).to(dev)
    kw = dict(denoising_steps=args.denoise, ensemble_size=args.ensemble, processing_res=0, match_input_res=True,
              show_progress_bar=False, color_map=None)

    def one(pipe, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return pipe(img, generator=g, **kw).depth_np

    streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
    for p, s in zip(pipes, streams):   # warm-up: programs built, kernels loaded, on the stream each pipeline will use
        with torch.cuda.stream(s):
            one(p, 0)
            one(p, 0)
    torch.cuda.synchronize()
    K = args.maps // N * N

    def sequential():
        outs = []
        with torch.cuda.stream(streams[0]):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                outs.append(one(pipes[0], 100 + k))
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3, outs

    def concurrent():
        outs = [None] * K
        errs = []

        def work(j):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[j]):
                    for k in range(j, K, N):
                        outs[k] = one(pipes[j], 100 + k)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(j,)) for j in range(N)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        if errs:
            raise errs[0]
        return (time.perf_counter() - t0) / K * 1e3, outs

    for r in range(args.rounds):
        ms_s, o_s = sequential()
        ms_c, o_c = concurrent()
        same = all(np.array_equal(a, b) for a, b in zip(o_s, o_c))
        print(f"round={r} E={args.ensemble} maps={K} sequential {ms_s:.2f} ms/map   in_flight={N} {ms_c:.2f} ms/map   "
              f"ratio {ms_s / ms_c:.4f}   bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
