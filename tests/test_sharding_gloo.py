"""The N > 1 path on CPU: two processes over gloo shard a batch, score their shards and gather
the fixed-size records to rank 0 with ONE collective.  The per-shard scorer here is the CPU
oracle (test infrastructure) standing in for the GPU scorer: what is under test is
pyascore_amd.shard -- partitioning, padding, the single gather, reassembly in input order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(summary, k):
    n = summary["best_score"].shape[0]
    cols = [summary["best_score"].view(np.int32)[:, None], summary["n_sig"].astype(np.int32)[:, None],
            summary["best_sig"].view(np.int32).reshape(n, 2), summary["ascores"].view(np.int32),
            summary["alt_mask"].view(np.int32).reshape(n, 2 * k)]
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def _skewed_job(n=480, seed=77):
    """A job whose per-PSM work spans three orders of magnitude and whose largest n_of_mod differs
    between shards: light PSMs (1 of 2 sites) first, then a mix, then heavy ones (4 of 12 sites)."""
    from pyascore_amd import synth
    desc = synth.describe("cfg3", n_psm=n, seed=seed)
    third = n // 3
    desc["n_mod"][:third], desc["n_sites"][:third], desc["L"][:third] = 1, 2, 10
    desc["n_mod"][-third:], desc["n_sites"][-third:], desc["L"][-third:] = 4, 12, 30
    return desc


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import harness, orc
    from pyascore_amd import shard, synth
    # the benchable path: every rank holds only the job DESCRIPTION, cuts the same partition from it
    # and generates the spectra of its own slice
    desc = _skewed_job()
    w = shard.work_estimate_shapes(desc["n_sites"], desc["n_mod"], desc["L"], desc["max_charge"])
    ranges = shard.partition(w, world)
    max_k = int(desc["n_mod"].max())
    lo, hi = ranges[rank]
    mine = synth.make_slice(desc, lo, hi, threads=1)
    assert int(mine["n_of_mod"].max()) <= max_k
    scorer = harness.make_scorer(orc.OracleAscore, desc["settings"], kind="oracle")
    calls = {"gather": 0}

    def gather_fn(t, dst):
        calls["gather"] += 1
        return shard.dist_gather(t, dst)

    # records packed at the JOB-wide width although rank 0's shard only holds n_of_mod = 1
    rec = torch.from_numpy(_records(scorer.score_batch(mine, max_k), max_k))
    out = shard.gather_records(rec, ranges, rank, max_k, gather_fn)
    assert calls["gather"] == 1                       # exactly one collective on the path
    # a rank that packs at its own shard's width is refused before it can enter the collective
    own_k = int(mine["n_of_mod"].max())
    if own_k < max_k:
        with pytest.raises(ValueError):
            shard.gather_records(torch.from_numpy(_records(scorer.score_batch(mine, own_k), own_k)), ranges, rank,
                                 max_k, gather_fn)
        assert calls["gather"] == 1
    # the whole-batch form (every rank holds the batch): same records
    full = synth.make_slice(desc, threads=1)
    rec2, ranges2 = shard.score_sharded(
        lambda sh, k: torch.from_numpy(_records(scorer.score_batch(sh, k), k)), full, rank, world, gather_fn)
    assert ranges2 == ranges and calls["gather"] == 2
    # the pipelined form bench.py uses: asynchronous gathers, waited for one batch later
    flights = [shard.dist_gather(torch.full((5, 3), 10 * step + rank, dtype=torch.int32), 0, async_op=True)
               for step in range(3)]
    for step, (work, parts) in enumerate(flights):
        work.wait()
        if rank == 0:
            assert [int(p[0, 0]) for p in parts] == [10 * step + r for r in range(world)]
        else:
            assert parts is None
    if rank == 0:
        assert torch.equal(out, rec2)
        np.save(out_path, out.numpy())
    else:
        assert out is None and rec2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_four_rank_gloo_gather_matches_single_process(tmp_path):
    from oracle import harness, orc
    from pyascore_amd import shard, synth
    from pyascore_amd.device import unpack_summary
    desc = _skewed_job()
    w = shard.work_estimate_shapes(desc["n_sites"], desc["n_mod"], desc["L"], desc["max_charge"])
    ranges = shard.partition(w, 4)
    per_rank = np.array([w[lo:hi].sum() for lo, hi in ranges])
    assert np.all(np.abs(per_rank - per_rank.mean()) <= 0.10 * per_rank.mean()), per_rank   # work-balanced
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) > 3 * min(sizes)                # ... which is far from an equal split by count
    out = str(tmp_path / "rec.npy")
    mp.spawn(_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    got = unpack_summary(np.load(out), 4)
    batch = synth.make_slice(desc, threads=1)
    want = harness.make_scorer(orc.OracleAscore, desc["settings"], kind="oracle").score_batch(batch, 4)
    for key in ("best_score", "n_sig", "best_sig", "ascores", "alt_mask"):
        assert np.array_equal(got[key], want[key]), key


def test_work_estimate_is_vectorised_and_counts_termini():
    from math import comb
    from pyascore_amd import shard, synth
    batch, _ = synth.make_batch("cfg3", n_psm=300, seed=5)
    w = shard.work_estimate(batch, "STY", 2)
    off = batch["pep_off"]
    for i in (0, 17, 299):
        pep = bytes(batch["pep"][off[i]:off[i + 1]]).decode()
        ns = sum(c in "STY" for c in pep)
        assert w[i] == max(1, comb(ns, int(batch["n_of_mod"][i]))) * (len(pep) - 1) * 2
    psms = [dict(mz=[100.5], intensity=[1.0], peptide="AKSTA", n_of_mod=1), dict(mz=[100.5], intensity=[1.0], peptide="SAAAK", n_of_mod=1)]
    assert list(shard.count_sites(synth.pack_batch(psms), "STnc")) == [4, 2]
    desc = synth.describe("cfg3", 1_000_000, seed=1)
    import time
    t = time.perf_counter()
    shard.work_estimate_shapes(desc["n_sites"], desc["n_mod"], desc["L"], desc["max_charge"])
    assert time.perf_counter() - t < 2.0


def _pipeline_worker(rank, world, port, out_path):
    """bench.py's N > 1 step loop (shard.StepPipeline) on CPU: three steps over gloo, asynchronous
    gathers one step behind, the job-wide record shape on every rank."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import harness, orc
    from pyascore_amd import shard, synth
    desc = _skewed_job(n=90)
    w = shard.work_estimate_shapes(desc["n_sites"], desc["n_mod"], desc["L"], desc["max_charge"])
    ranges = shard.partition(w, world)
    max_k = int(desc["n_mod"].max())
    lo, hi = ranges[rank]
    mine = synth.make_slice(desc, lo, hi, threads=1)
    scorer = harness.make_scorer(orc.OracleAscore, desc["settings"], kind="oracle")
    state = {"rec": None, "runs": 0, "gathers": 0}

    def run():
        state["rec"] = torch.from_numpy(_records(scorer.score_batch(mine, max_k), max_k))
        state["runs"] += 1

    def pack(out):
        assert tuple(out.shape) == (hi - lo, shard.record_width(max_k))
        out.copy_(state["rec"])

    def gather(t, dst):
        state["gathers"] += 1
        assert tuple(t.shape) == (max(h - l for l, h in ranges), shard.record_width(max_k))   # job-wide shape
        return shard.dist_gather(t, dst, async_op=True)

    pipe = shard.StepPipeline(run, pack, hi - lo, max(h - l for l, h in ranges), shard.record_width(max_k), "cpu",
                              gather_fn=gather)
    for step in range(3):
        pipe.step()
        assert len(pipe.in_flight) == 1                   # at most one gather behind
    pipe.drain()
    assert not pipe.in_flight and state["runs"] == 3 and state["gathers"] == 3 and pipe.steps == 3
    out = pipe.gathered(ranges)
    if rank == 0:
        np.save(out_path, out.numpy())
    else:
        assert out is None
    # a world of one without a process group behind it: the pipeline only runs the scorer
    solo = shard.StepPipeline(run, pack, hi - lo, hi - lo, shard.record_width(max_k), "cpu", enabled=False)
    solo.step()
    solo.drain()
    assert state["runs"] == 4 and state["gathers"] == 3
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_pipeline_is_what_bench_runs(tmp_path):
    from oracle import harness, orc
    from pyascore_amd import synth
    from pyascore_amd.device import unpack_summary
    import inspect
    import bench
    assert "shard.StepPipeline(" in inspect.getsource(bench.main)      # the bench's loop IS this class
    out = str(tmp_path / "rec.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    desc = _skewed_job(n=90)
    got = unpack_summary(np.load(out), 4)
    want = harness.make_scorer(orc.OracleAscore, desc["settings"], kind="oracle").score_batch(synth.make_slice(desc, threads=1), 4)
    for key in ("best_score", "n_sig", "best_sig", "ascores", "alt_mask"):
        assert np.array_equal(got[key], want[key]), key
