#!/usr/bin/env python
"""Tuning sweep (run on the MI355X): times every GEMM tile variant on the dominant UNet / VAE shapes
at the benchmark batch (E=10 members) and both flash-attention generations; prints a table and writes
gpurun_out/sweep.json.  Used to set mg_igemm_auto_variant() - not part of the product path."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O, weights as Wm  # noqa: E402

VARIANTS = tuple(int(v) for v in os.environ.get("SWEEP_VARIANTS", "0,33,34,35").split(","))
dev = torch.device("cuda:0")
L.init(0)


def timeit(fn, warm=2, iters=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_case(name, B, HW, Cin, N, taps, geglu=False, res=True):
    H = W = HW
    M = B * H * W
    K = taps * Cin
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    n_out = N // 2 if geglu else N
    out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, n_out, generator=g).to(dev, torch.bfloat16) if (res and not geglu) else None
    flops = 2.0 * M * N * K
    rows = {}
    ref = None
    rounds = int(os.environ.get("SWEEP_ROUNDS", "1"))   # > 1: variants interleaved round-robin, median reported
    opsv, times = {}, {v: [] for v in VARIANTS}
    for v in VARIANTS:
        opsv[v] = O.igemm(x, w, out, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, N=N, taps=taps, stride=1,
                          pad=1 if taps == 9 else 0, bias=bias, residual=r,
                          epi=L.EPI_GEGLU if geglu else L.EPI_BF16, variant=v)
    for rnd in range(rounds):
        for v in VARIANTS:
            if times[v] is None:
                continue
            try:
                times[v].append(timeit(lambda: O.launch(opsv[v]), warm=2 if rnd == 0 else 1, iters=8 if rounds == 1 else 5))
            except Exception as e:  # noqa: BLE001
                times[v] = None
                print(f"  {name} v{v}: {e}")
                continue
            if rnd == 0:
                o = out.float()
                if ref is None:
                    ref = o.clone()
                times[v].append(-float((o - ref).abs().max()))   # stash the error as a negative entry
    for v in VARIANTS:
        if times[v] is None:
            rows[v] = None
            continue
        err = -min(times[v])
        ts = sorted(t for t in times[v] if t > 0)
        ms = ts[len(ts) // 2]
        rows[v] = (ms, flops / ms / 1e9, err)
    best = max((k for k in rows if rows[k]), key=lambda k: rows[k][1])
    print(f"{name:34s} M={M:8d} N={N:5d} K={K:6d} | " + " ".join(
        f"v{k}:{rows[k][1]:6.0f}" if rows[k] else f"v{k}:  fail" for k in rows) + f" | best v{best}"
        + "".join(f" !!v{k} differs {rows[k][2]:.3g}" for k in rows if rows[k] and rows[k][2] > 0.05 and (k < 40 or k >= 60)))
    return {"name": name, "M": M, "N": N, "K": K, "tflops": {str(k): (rows[k][1] if rows[k] else None) for k in rows},
            "best": best}


def flash_case(B, heads, T):
    C = heads * 64
    g = torch.Generator(device="cpu").manual_seed(2)
    qkv = torch.randn(B, T, 2 * C, generator=g).to(dev, torch.bfloat16)
    ldvt = (T + 63) // 64 * 64
    vt = torch.randn(B, C, ldvt, generator=g).to(dev, torch.bfloat16)
    out = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    flops = 4.0 * B * heads * T * T * 64
    res = {}
    ref = None
    for v in (1, 2, 0, 3, 4, 5, 6, 7):
        op = O.flash_attn64(qkv, qkv.data_ptr() + C * 2, vt, out, B=B, heads=heads, Ntok=T, ldq=2 * C, ldo=C,
                            ldvt=ldvt, sq=T * 2 * C, sk=T * 2 * C, svt=C * ldvt, so=T * C, scale=0.125, variant=v)
        ms = timeit(lambda: O.launch(op), iters=4)
        o = out.float()
        if ref is None:
            ref = o.clone()
        res[v] = (ms, flops / ms / 1e9, float((o - ref).abs().max()))
    names = {1: "gen1", 2: "burst", 0: "split(default)", 3: "8w", 4: "b128/shfl", 5: "b128/permlane", 6: "8w+permlane",
             7: "8w+shfl"}
    print(f"flash B={B} heads={heads} T={T}: " + "  ".join(f"{names[v]} {res[v][1]:.0f}" for v in res)
          + "  | max diff vs gen1: " + " ".join(f"{res[v][2]:.2g}" for v in res))
    return {"B": B, "heads": heads, "T": T, "tflops": {names[v]: res[v][1] for v in res}}


def main():
    E = 10
    out = {"gemm": [], "flash": []}
    cases = [
        ("unet.conv 320->320 @96", E, 96, 320, 320, 9),
        ("unet.conv 640->320 @96", E, 96, 640, 320, 9),
        ("unet.conv 960->320 @96", E, 96, 960, 320, 9),
        ("unet.conv 640->640 @48", E, 48, 640, 640, 9),
        ("unet.conv 1280->640 @48", E, 48, 1280, 640, 9),
        ("unet.conv 1280->1280 @24", E, 24, 1280, 1280, 9),
        ("unet.conv 2560->1280 @24", E, 24, 2560, 1280, 9),
        ("unet.conv 1280->1280 @12", E, 12, 1280, 1280, 9),
        ("unet.linear 320->320 @96", E, 96, 320, 320, 1),
        ("unet.linear 320->960 @96 (qkv)", E, 96, 320, 960, 1),
        ("unet.linear 640->1920 @48 (qkv)", E, 48, 640, 1920, 1),
        ("unet.linear 1280->320 @96", E, 96, 1280, 320, 1),
        ("unet.linear 960->320 @96", E, 96, 960, 320, 1),
        ("unet.linear 5120->1280 @24", E, 24, 5120, 1280, 1),
        ("unet.linear 640->640 @48", E, 48, 640, 640, 1),
        ("unet.linear 2560->640 @48", E, 48, 2560, 640, 1),
        ("unet.linear 1280->1280 @24", E, 24, 1280, 1280, 1),
        ("vae.conv 512->512 @96 B10", E, 96, 512, 512, 9),
        ("vae.conv 512->512 @192 B2", 2, 192, 512, 512, 9),
        ("vae.conv 256->256 @384 B2", 2, 384, 256, 256, 9),
        ("vae.conv 128->128 @768 B2", 2, 768, 128, 128, 9),
    ]
    only = os.environ.get("SWEEP_ONLY")   # substring filter on the case names
    for c in cases:
        if only is None or only in c[0]:
            out["gemm"].append(conv_case(*c))
    for c in [("unet.geglu 320->2560 @96", E, 96, 320, 2560, 1), ("unet.geglu 640->5120 @48", E, 48, 640, 5120, 1),
              ("unet.geglu 1280->10240 @24", E, 24, 1280, 10240, 1)]:
        out["gemm"].append(conv_case(*c, geglu=True))
    if not os.environ.get("SWEEP_NO_FLASH"):
        for c in [(E, 5, 9216), (E, 10, 2304), (E, 20, 576), (E, 20, 144)]:
            out["flash"].append(flash_case(*c))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
