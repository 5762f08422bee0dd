"""Member parallelism: ensemble members are independent from noise init to decoded map
(marigold_depth_pipeline.py:281-289), so they shard over the GPUs of one node - one process per
GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests) - and
meet in ONE gather of [E/G, C, H, W] predictions before the per-pixel aggregation (:294-300).
The payload is small (2.4 MB per fp32 depth member at 768^2), i.e. latency-bound on xGMI: a
direct gather to the aggregating rank (each peer's slab travels over its own link) instead of a
ring.  The reference has no multi-GPU path.
"""
import torch
import torch.distributed as dist


def is_on(group=None):
    return dist.is_available() and dist.is_initialized()


def world_size(group=None):
    return dist.get_world_size(group) if is_on(group) else 1


def rank(group=None):
    return dist.get_rank(group) if is_on(group) else 0


def shard_members(E, world, r, offset=0):
    """Members owned by rank r: e with (e + offset) % world == r (round-robin => balanced)."""
    return [e for e in range(E) if (e + offset) % world == r]


def members_per_rank(E, world, offset=0):
    return [shard_members(E, world, r, offset) for r in range(world)]


def gather_members(local, E, chw, device, group=None, root=None, offset=0, dtype=torch.float32, force=False):
    """Collect every rank's member predictions into [E, C, H, W] in member order.

    root=None -> all ranks receive the full stack (all_gather); root=k -> only rank k does
    (returns None elsewhere).  Uneven shards are padded to ceil(E / world) rows for the collective.
    ``force``: run the collective even in a group of ONE rank (a gather to self) - how the RCCL code path is
    executed and tested on a single GPU (bench.py MARIGOLD_BENCH_FORCE_DIST=1).
    """
    G, r = world_size(group), rank(group)
    if G == 1 and not (force and is_on(group)):
        return local
    per = (E + G - 1) // G
    out_device = device
    if dist.get_backend(group) == "gloo":   # CPU tests / debugging: gloo collectives take host tensors
        device = torch.device("cpu")
    buf = torch.zeros((per,) + tuple(chw), device=device, dtype=dtype)
    if local is not None and local.shape[0] > 0:
        buf[:local.shape[0]].copy_(local)
    if root is None:
        parts = [torch.empty_like(buf) for _ in range(G)]
        dist.all_gather(parts, buf, group=group)
    else:
        parts = [torch.empty_like(buf) for _ in range(G)] if r == root else None
        dist.gather(buf, parts, dst=root, group=group)
        if r != root:
            return None
    out = torch.empty((E,) + tuple(chw), device=device, dtype=dtype)
    for rr, idx in enumerate(members_per_rank(E, G, offset)):
        if idx:
            out[idx] = parts[rr][:len(idx)]
    return out.to(out_device)
