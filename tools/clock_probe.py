#!/usr/bin/env python
"""What the chip clocks at under each of the dominant kernels (sysfs sclk / power sampled at ~1 kHz while one launch is
repeated for ~1.5 s): the MFMA roof a kernel can be priced against is 2.5 PFLOP/s x sclk / 2.4 GHz.  Tuning tool."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GpuTelemetry  # noqa: E402
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)


def probe(name, op, flops, seconds=1.2):
    O.launch(op)
    torch.cuda.synchronize()
    tel = GpuTelemetry(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    tel.start(period=0.001)
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            O.launch(op)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    st = tel.stop() or {}
    ms = e0.elapsed_time(e1) / n
    clk = (st.get("sclk_mhz") or {}).get("mean")
    tf = flops / ms / 1e9
    roof = 2500.0 * clk / 2400.0 if clk else float("nan")
    print(f"{name:44s} {ms * 1e3:8.1f} us {tf:7.0f} TF/s | sclk mean {clk} min {(st.get('sclk_mhz') or {}).get('min')} MHz, "
          f"power {(st.get('power_w') or {}).get('mean')} W | {tf / roof:.3f} of the roof at that clock ({roof:.0f} TF/s)", flush=True)


def flash(B, heads, T, variant):
    C = heads * 64
    g = torch.Generator(device="cpu").manual_seed(2)
    qkv = torch.randn(B, T, 3 * C, generator=g).to(dev, torch.bfloat16)
    vt = qkv[:, :, 2 * C:].permute(0, 2, 1).contiguous()
    out = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    op = O.flash_attn64(qkv, qkv[:, :, C:], vt, out, B=B, heads=heads, Ntok=T, ldq=3 * C, ldo=C, ldvt=T,
                        sq=T * 3 * C, sk=T * 3 * C, svt=C * T, so=T * C, scale=0.125, variant=variant)
    probe(f"flash v{variant} B{B} h{heads} T{T}", op, 4.0 * B * heads * T * T * 64)
    return qkv, vt, out


def gemm(name, M, N, K, variant=0, taps=1, HW=None, B=1, geglu=False):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(M, K // taps, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, torch.bfloat16)
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
    if taps == 1:
        op = O.linear(x, w, out, M=M, K=K, N=N, variant=variant, epi=L.EPI_GEGLU if geglu else L.EPI_BF16)
    else:
        op = O.igemm(x, w, out, B=B, H=HW, W=HW, Cin=K // taps, Ho=HW, Wo=HW, N=N, taps=taps, stride=1, pad=1, variant=variant)
    probe(name, op, 2.0 * M * N * K)
    return x, w, out


def conv_patch(name, B, HW, Cin, N):
    g = torch.Generator(device="cpu").manual_seed(1)
    M = B * HW * HW
    x = (torch.randn(M, Cin, generator=g) * 0.5).to(dev, torch.bfloat16)
    w = (torch.randn(N, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(dev, torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    op = O.conv3x3(x, w, out, B=B, H=HW, W=HW, C0=Cin, N=N)
    probe(name, op, 2.0 * M * N * 9 * Cin)
    return x, w, out


if __name__ == "__main__":
    vs = [int(v) for v in os.environ.get("FLASH_VARIANTS", "6,13,14").split(",")]
    for v in vs:
        flash(10, 5, 9216, v)
    gemm("gemm 4096^3 (auto tile)", 4096, 4096, 4096)
    gemm("gemm 8192^3 (auto tile)", 8192, 8192, 8192)
    gemm("geglu 320->2560 @96^2 B10", 92160, 2560, 320, geglu=True)
    gemm("linear 320->320 @96^2 B10", 92160, 320, 320)
    conv_patch("conv_patch 320->320 @96^2 B10", 10, 96, 320, 320)
    conv_patch("conv_patch 512->512 @192^2 B2", 2, 192, 512, 512)
    gemm("igemm conv 1280->1280 @24^2 B10", 10 * 576, 1280, 9 * 1280, taps=9, HW=24, B=10)
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    tel = GpuTelemetry(0)
    tel.start(period=0.001)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(10):
            b.copy_(a)
        torch.cuda.synchronize()
    print("copy 1 GiB loop:", tel.stop())
