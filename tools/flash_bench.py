#!/usr/bin/env python
"""Flash-attention variants side by side on the MI355X (interleaved rounds, median): TFLOP/s at the UNet's self-attention
shapes and the error of each against CPU SDPA in fp32 on a sub-sampled set of queries.  Tuning tool, not product path."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

VARIANTS = tuple(int(v) for v in os.environ.get("FLASH_VARIANTS", "25,26,27,126,127").split(","))
ROUNDS = int(os.environ.get("FLASH_ROUNDS", "5"))
dev = torch.device("cuda:0")
L.init(0)
WS = torch.zeros(O.FLASH_WS_BYTES, dtype=torch.uint8, device=dev)
HP = (26, 27, 126, 127)   # the hand-placed stream on 32x32x16 / 16x16x32 MFMAs; + 100: with the key-split workspace


def timeit(fn, warm=1, iters=4):
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


def case(B, heads, T, qscale=1.0):
    global VARIANTS
    all_variants = VARIANTS
    VARIANTS = tuple(v for v in VARIANTS if v not in HP or (T % 256 == 0 and T >= 256))   # the hand-placed form's shapes
    try:
        _case(B, heads, T, qscale)
    finally:
        VARIANTS = all_variants


def _case(B, heads, T, qscale=1.0):
    C = heads * 64
    g = torch.Generator(device="cpu").manual_seed(2)
    qkv = (torch.randn(B, T, 3 * C, generator=g)).to(torch.bfloat16)
    qkv[:, :, :C] *= qscale
    ldvt = (T + 63) // 64 * 64
    qkd = qkv.to(dev)
    vt = torch.zeros(B, C, ldvt, device=dev, dtype=torch.bfloat16)
    vt[:, :, :T] = qkd[:, :, 2 * C:].permute(0, 2, 1)
    vtp = O.permute_vt_keys(vt)
    flops = 4.0 * B * heads * T * T * 64
    # reference on image 0, head 0, every 37th query
    q, k, v = (qkv[0, :, i * C:i * C + 64].float() for i in range(3))
    qs = q[::37]
    ref = F.scaled_dot_product_attention(qs[None, None], k[None, None], v[None, None])[0, 0]
    outs, ops, times = {}, {}, {v: [] for v in VARIANTS}
    for v in VARIANTS:
        outs[v] = torch.full((B, T, C), float("nan"), device=dev, dtype=torch.bfloat16)
        vv = v - 100 if v > 100 else v
        ops[v] = O.flash_attn64(qkd, qkd[:, :, C:], vtp if (13 <= v <= 20 or v in (22, 23, 25) or v in HP) else vt, outs[v], B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=ldvt,
                                sq=T * 3 * C, sk=T * 3 * C, svt=C * ldvt, so=T * C, scale=0.125, variant=vv, vt_perm=(v in (19, 20, 25) or v in HP),
                                ws=WS if v > 100 else None, ws_bytes=O.FLASH_WS_BYTES if v > 100 else 0)
    for rnd in range(ROUNDS):
        for v in VARIANTS:
            times[v].append(timeit(lambda: O.launch(ops[v])))
    line = f"flash B={B} heads={heads} T={T} qscale={qscale}:"
    if os.environ.get("FLASH_DBG"):   # generation 3 only: shader cycles / wall ticks per workgroup
        for v in VARIANTS:
            if v in HP:   # the hand-placed form: cycles of the key loop per wave
                nwg = (T // 256) * heads * B + 1024
                dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
                op = O.flash_attn64(qkd, qkd[:, :, C:], vtp, outs[v], B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=ldvt, sq=T * 3 * C,
                                    sk=T * 3 * C, svt=C * ldvt, so=T * C, scale=0.125, variant=v - 100 if v > 100 else v, vt_perm=True, dbg=dbg,
                                    ws=WS if v > 100 else None, ws_bytes=O.FLASH_WS_BYTES if v > 100 else 0)
                O.launch(op)
                O.launch(op)
                torch.cuda.synchronize()
                d = dbg.view(nwg, 4, 8).cpu()
                used = d[:, 0, 0] > 0
                st = d[used][:, 0, 2:8].double() / 100.0      # us: entry, DMA + Q issued, first data landed, stream in, stream out, stored
                ph = st[:, 1:] - st[:, :-1]
                print(f"   v{v} phases (us, median over workgroups): issue DMA + queries {ph[:, 0].median():.2f}, wait first data {ph[:, 1].median():.2f}, "
                      f"first scores / reference {ph[:, 2].median():.2f}, stream {ph[:, 3].median():.2f}, store {ph[:, 4].median():.2f}; "
                      f"kernel span (first entry -> last store) {float(st[:, 5].max() - st[:, 0].min()):.1f} us", flush=True)
                cyc = d[used][:, :, 0].double().flatten()
                tick = (d[used][:, :, 1] & ((1 << 40) - 1)).double().flatten()
                tiles = (d[used][:, :, 1] >> 40).double().flatten()
                per = cyc / tiles
                print(f"   v{v}: {int(used.sum())} workgroups; last segment {tiles.min():.0f}-{tiles.max():.0f} tiles; key loop {per.median():.0f} cycles per 64-key "
                      f"iteration (32 MFMAs = 1024), min {per.min():.0f} max {per.max():.0f}; clock {float((cyc / tick).median()) * 100:.0f} MHz; "
                      f"loop time per workgroup median {float((tick / 100).median()):.1f} us max {float((tick / 100).max()):.1f} us", flush=True)
                continue
            if v < 9:
                continue
            if v > 16:
                continue
            nw = 8 if v in (9, 11, 13, 15) else 4
            nwg = -(-T // (nw * 32)) * heads * B
            dbg = torch.zeros(nwg * nw * 8, dtype=torch.int64, device=dev)
            op = O.flash_attn64(qkd, qkd[:, :, C:], vtp if (13 <= v <= 20 or v in (22, 23, 25, 26)) else vt, outs[v], B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=ldvt,
                                sq=T * 3 * C, sk=T * 3 * C, svt=C * ldvt, so=T * C, scale=0.125, variant=v, vt_perm=(v in (19, 20, 25, 26)), dbg=dbg)
            O.launch(op)
            O.launch(op)
            torch.cuda.synchronize()
            d = dbg.view(nwg, nw, 8).double().cpu()
            nt = (-(-T // 64) + 3) // 4 * 4
            cyc, tick = d[:, 0, 0], d[:, 0, 1]
            per = d[:, :, 2:5].mean(dim=(0, 1)) / nt
            wv = d[:, :, 2:5].mean(dim=0) / nt          # per wave index
            print(f"   v{v}: {nwg} workgroups x {nw} waves, cycles per workgroup median {cyc.median():.0f} = {cyc.median() / nt:.0f} per tile; "
                  f"clock {float((cyc / tick).median()) * 100:.0f} MHz; per tile: barrier wait {per[0]:.0f}, phase A {per[1]:.0f}, phase B {per[2]:.0f}; "
                  f"rescales per wave {d[:, :, 5].mean():.1f} of {nt}", flush=True)
            print("        per wave index (barrier / A / B): " + "  ".join(f"{wv[w, 0]:.0f}/{wv[w, 1]:.0f}/{wv[w, 2]:.0f}" for w in range(nw)), flush=True)
    for v in VARIANTS:
        ts = sorted(times[v])
        ms = ts[len(ts) // 2]
        got = outs[v][0, ::37, :64].float().cpu()
        err = float((got - ref).abs().max())
        nan = int(torch.isnan(outs[v].float()).sum())
        line += f"  v{v}: {flops / ms / 1e9:6.0f} TF/s ({ms * 1e3:7.1f} us, min {ts[0] * 1e3:7.1f}) err {err:.2e}" + (f" NAN {nan}" if nan else "")
    print(line, flush=True)


if __name__ == "__main__":
    if os.environ.get("FLASH_OCC"):   # occupancy steps: B x heads x 72 workgroups of 4 waves at 9 216 tokens
        for bh in [int(x) for x in os.environ.get('FLASH_OCC_BH', '1,2,3,5,7,8,10,11,14,15,21,22').split(',')]:
            case(1, bh, 9216)
        sys.exit(0)
    E = int(os.environ.get("FLASH_E", "10"))
    case(E, 5, 9216)
    case(E, 10, 2304)
    case(E, 20, 576)
    case(E, 20, 144)
    case(1, 5, 9216)
    case(2, 5, 9216, qscale=4.0)
