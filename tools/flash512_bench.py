#!/usr/bin/env python
"""MG_OP_FLASH_ATTN512 (the VAE mid-block attention as one flash kernel) against the three-stage form (scores GEMM -> row softmax
-> P V GEMM) at the decoder's shape, interleaved rounds on one MI355X.  Tuning tool, not product path."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marigold_amd import _lib as L, ops as O  # noqa: E402

dev = torch.device("cuda:0")
L.init(0)
ROUNDS = int(os.environ.get("FLASH_ROUNDS", "5"))


def timeit(fn, warm=1, iters=3):
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


def case(B, T):
    C = 512
    g = torch.Generator().manual_seed(T)
    qk = (torch.randn(B * T + 8, 2 * C, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(B, T, C, generator=g).to(torch.bfloat16)
    ldp = (T + 63) // 64 * 64
    Tn = (T + 7) // 8 * 8
    qkd = qk.to(dev)
    vt = torch.zeros(B, C, ldp, device=dev, dtype=torch.bfloat16)
    vt[:, :, :T] = v.to(dev).permute(0, 2, 1)
    o1 = torch.full((B, T, C), float("nan"), device=dev, dtype=torch.bfloat16)
    o2 = torch.full((B, T, C), float("nan"), device=dev, dtype=torch.bfloat16)
    s = torch.empty(B * T * ldp, device=dev, dtype=torch.float32)
    p = torch.empty(B * T * ldp, device=dev, dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(C)
    flash = O.flash_attn512(qkd, qkd.data_ptr() + C * 2, vt, o1, B=B, Ntok=T, ldq=2 * C, ldo=C, ldvt=ldp,
                            sq=T * 2 * C, sk=T * 2 * C, svt=C * ldp, so=T * C, scale=scale)
    three = [O.igemm(qkd, qkd.data_ptr() + C * 2, s, B=1, H=T, W=1, Cin=C, Ho=T, Wo=1, N=Tn, epi=L.EPI_F32, ldo=ldp, lda=2 * C,
                     ldw=2 * C, batch_z=B, n_alg=T, zstrides=(T * 2 * C, T * 2 * C, T * ldp, 0), scale=scale),
             O.softmax_rows(s, p, R=B * T, ncols=T, lds=ldp, ldp=ldp),
             O.igemm(p, vt, o2, B=1, H=T, W=1, Cin=ldp, Ho=T, Wo=1, N=C, lda=ldp, ldw=ldp, batch_z=B,
                     zstrides=(T * ldp, C * ldp, T * C, 0))]
    t1, t3 = [], []
    for _ in range(ROUNDS):
        t1.append(timeit(lambda: O.launch(flash)))
        t3.append(timeit(lambda: [O.launch(x) for x in three]))
    t1.sort(); t3.sort()
    m1, m3 = t1[len(t1) // 2], t3[len(t3) // 2]
    flops = 4.0 * B * T * T * C
    rows = slice(0, T, 41)
    q0, k0 = qk[:T, :C].float(), qk[:T, C:].float()
    ref = F.scaled_dot_product_attention(q0[rows][None, None], k0[None, None], v[0].float()[None, None])[0, 0]
    e1 = (o1[0, rows].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    e3 = (o2[0, rows].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"vae attention B={B} T={T}: flash512 {m1:.3f} ms ({flops / m1 * 1e-9:.0f} TFLOP/s, rel err {e1:.2e}) | "
          f"three-stage {m3:.3f} ms ({flops / m3 * 1e-9:.0f} TFLOP/s, rel err {e3:.2e}) | x{m3 / m1:.2f}", flush=True)


if __name__ == "__main__":
    for B, T in ((1, 9216), (10, 9216), (10, 2304), (1, 2304)):
        case(B, T)
