"""PSM sharding across the GPUs of one node.

PSMs are independent, so the batch is cut into contiguous ranges, one per rank (one process
per GPU), balanced by the estimated work  C(n_sites, n_mods) x (L - 1) x ion types x charges.
There is no collective on the data path; the only exchange is ONE gather of fixed-size summary
records to rank 0 at the end (RCCL when the tensors are on the GPU, gloo in the CPU tests).
"""
import numpy as np

from .synth import slice_batch

_COMB = None


def comb_table():
    """C(n, k) for n, k < 65 as float64 (exact up to 2^53; only used as a weight)."""
    global _COMB
    if _COMB is None:
        t = np.zeros((65, 65), np.float64)
        t[:, 0] = 1.0
        for n in range(1, 65):
            t[n, 1:] = t[n - 1, 1:] + t[n - 1, :-1]
        _COMB = t
    return _COMB


def work_estimate_shapes(n_sites, n_mod, L, max_charge, n_types=2):
    """Per-PSM work  max(1, C(n_sites, n_mod)) x (L - 1) x ion types x charges  from shape arrays
    (SURVEY.md section 8(e)); vectorised, so a million PSMs cost milliseconds."""
    n_sites = np.clip(np.asarray(n_sites, np.int64), 0, 64)
    k = np.asarray(n_mod, np.int64)
    c = np.where((k >= 0) & (k <= n_sites), comb_table()[n_sites, np.clip(k, 0, 64)], 0.0)
    return (np.maximum(c, 1.0) * np.maximum(np.asarray(L, np.int64) - 1, 1) * float(n_types) *
            np.maximum(np.asarray(max_charge, np.int64), 1))


def count_sites(batch, mod_group="STY"):
    """Modifiable residues per PSM of a CSR batch ('n' / 'c' in the group admit the termini)."""
    pep, off = np.asarray(batch["pep"]), np.asarray(batch["pep_off"], np.int64)
    letters = np.frombuffer(mod_group.replace("n", "").replace("c", "").encode(), dtype=np.uint8)
    is_site = np.isin(pep, letters)
    nonempty = off[1:] > off[:-1]
    if "n" in mod_group:
        is_site[off[:-1][nonempty]] = True
    if "c" in mod_group:
        is_site[off[1:][nonempty] - 1] = True
    csum = np.concatenate([[0], np.cumsum(is_site)])
    return csum[off[1:]] - csum[off[:-1]]


def work_estimate(batch, mod_group="STY", n_types=2):
    """Per-PSM work estimate of a CSR batch used for balancing."""
    off = np.asarray(batch["pep_off"], np.int64)
    return work_estimate_shapes(count_sites(batch, mod_group), batch["n_of_mod"], off[1:] - off[:-1],
                                batch["max_charge"], n_types)


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


def record_width(max_k):
    """int32 words of one gathered summary record (device.DevicePlan.packed_summary)."""
    return 4 + 3 * int(max_k)


def score_sharded(score_fn, batch, rank, world_size, gather_fn, mod_group="STY", n_types=2):
    """Scores rank's shard with ``score_fn(shard_batch, max_k) -> int32 [n, 4 + 3 * max_k] records``
    (a torch tensor on whatever device the process group communicates on) and gathers all records
    to rank 0 with ONE call of ``gather_fn(tensor, dst=0) -> list of tensors or None``.

    ``max_k`` is the JOB-wide largest n_of_mod: every rank packs at that width, whatever its own
    shard holds (an empty shard included), so the fixed-size gather sees equal shapes on all ranks.
    Returns (records of the whole batch in input order, ranges) on rank 0, (None, ranges)
    elsewhere.  Shards are padded to the largest shard so one fixed-size gather suffices."""
    ranges = partition(work_estimate(batch, mod_group, n_types), world_size)
    max_k = max(1, int(np.max(batch["n_of_mod"]))) if int(batch["n_psm"]) else 1
    lo, hi = ranges[rank]
    rec = score_fn(slice_batch(batch, lo, hi), max_k)
    return gather_records(rec, ranges, rank, max_k, gather_fn), ranges


def gather_records(rec, ranges, rank, max_k, gather_fn):
    """The single collective of the path: pads this rank's records to the longest shard and gathers
    them to rank 0, which returns them in input order (None elsewhere)."""
    import torch
    lo, hi = ranges[rank]
    width = record_width(max_k)
    if tuple(rec.shape) != (hi - lo, width):
        raise ValueError("rank %d packed records of shape %s; the job-wide shape is (%d, %d)"
                         % (rank, tuple(rec.shape), hi - lo, width))
    longest = max(h - l for l, h in ranges)
    padded = torch.zeros((longest, width), dtype=rec.dtype, device=rec.device)
    padded[: hi - lo] = rec
    parts = gather_fn(padded, 0)
    if rank != 0:
        return None
    return torch.cat([parts[r][: h - l] for r, (l, h) in enumerate(ranges)], dim=0)


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


class StepPipeline:
    """The per-step loop of a rank in a multi-GPU job -- the code bench.py runs on 8 GPUs and the gloo
    test runs on CPU: score this rank's shard, pack its fixed-size records into one of two send
    buffers of the JOB-wide shape (longest shard x record_width(job max_k): equal on every rank), start
    the step's single gather asynchronously, and wait for the gather of the step before -- so the
    collective of batch i overlaps the kernels of batch i + 1 and at most one gather is in flight when
    the next one starts.

    run_fn()            enqueues / performs the scoring of the shard
    pack_fn(out)        writes the shard's records into out ([n_local, width] int32 view of the send buffer)
    gather_fn(t, dst, async_op=True) -> (work, parts): dist_gather by default

    Collects what a first real scaling run needs to explain itself: seconds spent waiting for
    gathers (gather_wait_s), steps, and on rank 0 the records of the last completed gather."""

    def __init__(self, run_fn, pack_fn, n_local, longest, width, device, gather_fn=None, enabled=True):
        import torch
        self.run_fn, self.pack_fn = run_fn, pack_fn
        self.n_local, self.longest, self.width = int(n_local), int(longest), int(width)
        self.gather_fn = gather_fn or (lambda t, dst: dist_gather(t, dst, async_op=True))
        self.enabled = enabled
        self.send = [torch.zeros((self.longest, self.width), dtype=torch.int32, device=device) for _ in range(2)] if enabled else []
        self.flip = 0
        self.in_flight = []
        self.gather_wait_s = 0.0
        self.steps = 0
        self.last_parts = None

    def _wait_oldest(self):
        import time
        work, parts = self.in_flight.pop(0)
        t = time.perf_counter()
        work.wait()
        self.gather_wait_s += time.perf_counter() - t
        self.last_parts = parts

    def step(self):
        self.run_fn()
        self.steps += 1
        if not self.enabled:
            return
        buf = self.send[self.flip]
        self.flip ^= 1
        self.pack_fn(buf[: self.n_local])
        self.in_flight.append(self.gather_fn(buf, 0))
        if len(self.in_flight) > 1:
            self._wait_oldest()

    def drain(self):
        while self.in_flight:
            self._wait_oldest()

    def reset_stats(self):
        self.gather_wait_s = 0.0
        self.steps = 0

    def gathered(self, ranges):
        """Rank 0, after drain(): the records of the last step of the whole job in input order."""
        import torch
        if self.last_parts is None:
            return None
        return torch.cat([self.last_parts[r][: h - l] for r, (l, h) in enumerate(ranges)], dim=0)
