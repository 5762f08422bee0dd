#!/usr/bin/env python
"""Does the 256 MB memory-side cache (MALL) give more than HBM bandwidth to a working set that fits?  Times (a) a copy y <- x
repeated back to back and (b) "write then read" pairs - a fill of a buffer followed by a read of it (the producer -> consumer
pattern of two consecutive launches) - for working sets from 16 MB to 2 GB.  torch kernels: a probe, not the product path."""
import torch

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)


def timed(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 2 // 2          # two bf16 buffers of mb / 2 each: the working set is mb
    x = torch.randn(n, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    iters = max(5, 4096 // mb)
    t_copy = timed(lambda: y.copy_(x), iters)
    t_read = timed(lambda: x.sum(dtype=torch.float32), iters)
    def pair():
        y.copy_(x)                         # producer: writes y (reads x)
        y.sum(dtype=torch.float32)         # consumer: reads y right away
    t_pair = timed(pair, iters)
    print(f"working set {mb:5d} MB: copy {2 * x.numel() * 2 / t_copy / 1e12:5.2f} TB/s (r+w)   read-only {x.numel() * 2 / t_read / 1e12:5.2f} TB/s"
          f"   copy+read pair {3 * x.numel() * 2 / t_pair / 1e12:5.2f} TB/s", flush=True)
    del x, y
