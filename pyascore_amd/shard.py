"""PSM sharding across the GPUs of one node.

PSMs are independent, so the batch is cut into contiguous ranges, one per rank (one process
per GPU), balanced by the estimated work  C(n_sites, n_mods) x (L - 1) x ion types x charges.
There is no collective on the data path; the only exchange is ONE gather of fixed-size summary
records to rank 0 at the end (RCCL when the tensors are on the GPU, gloo in the CPU tests).
"""
from math import comb

import numpy as np

from .synth import slice_batch


def work_estimate(batch, mod_group="STY", n_types=2):
    """Per-PSM work estimate used for balancing."""
    n = int(batch["n_psm"])
    pep, off = batch["pep"], batch["pep_off"]
    is_site = np.isin(pep, np.frombuffer(mod_group.encode(), dtype=np.uint8))
    csum = np.concatenate([[0], np.cumsum(is_site)])
    n_sites = csum[off[1:]] - csum[off[:-1]]
    L = (off[1:] - off[:-1]).astype(np.int64)
    k = batch["n_of_mod"].astype(np.int64)
    table = {}
    w = np.empty(n, np.float64)
    for i in range(n):
        key = (int(n_sites[i]), int(k[i]))
        if key not in table:
            table[key] = comb(*key) if key[1] <= key[0] else 0
        w[i] = max(1, table[key]) * max(1, L[i] - 1) * n_types * max(1, int(batch["max_charge"][i]))
    return w


def partition(weights, world_size):
    """Contiguous ranges [lo, hi) per rank with near-equal total weight."""
    n = len(weights)
    csum = np.concatenate([[0.0], np.cumsum(weights)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        bounds.append(int(np.searchsorted(csum, target, side="left")))
    bounds.append(n)
    for r in range(1, len(bounds)):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def score_sharded(score_fn, batch, rank, world_size, gather_fn, mod_group="STY", n_types=2):
    """Scores rank's shard with ``score_fn(shard_batch) -> int32 [n, width] records`` (a torch
    tensor on whatever device the process group communicates on) and gathers all records to
    rank 0 with ONE call of ``gather_fn(tensor, dst=0) -> list of tensors or None``.

    Returns (records of the whole batch in input order, ranges) on rank 0, (None, ranges)
    elsewhere.  Shards are padded to the largest shard so one fixed-size gather suffices."""
    import torch
    ranges = partition(work_estimate(batch, mod_group, n_types), world_size)
    lo, hi = ranges[rank]
    shard = slice_batch(batch, lo, hi)
    rec = score_fn(shard)
    longest = max(h - l for l, h in ranges)
    width = rec.shape[1]
    padded = torch.zeros((longest, width), dtype=rec.dtype, device=rec.device)
    padded[: hi - lo] = rec
    parts = gather_fn(padded, 0)
    if rank != 0:
        return None, ranges
    out = torch.cat([parts[r][: h - l] for r, (l, h) in enumerate(ranges)], dim=0)
    return out, ranges


def dist_gather(tensor, dst=0, async_op=False):
    """One torch.distributed.gather (RCCL on GPU tensors, gloo on CPU tensors).

    With async_op=True returns (work, parts): the collective runs on the backend's own stream and
    the caller's stream only waits when work.wait() is called -- a pipeline of batches calls it
    after enqueuing the next batch's kernels, so the gather of batch i overlaps the scoring of
    batch i+1 (the gathered tensor is a packed copy, the scorer's own buffers are free again)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    parts = [torch.empty_like(tensor) for _ in range(world)] if dist.get_rank() == dst else None
    work = dist.gather(tensor, parts, dst=dst, async_op=async_op)
    return (work, parts) if async_op else parts
