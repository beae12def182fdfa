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


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import harness, orc
    from pyascore_amd import shard, synth
    batch, settings = synth.make_batch("cfg3", n_psm=240, seed=77)
    k = int(batch["n_of_mod"].max())
    scorer = harness.make_scorer(orc.OracleAscore, settings, kind="oracle")
    calls = {"gather": 0}

    def score_fn(sh):
        return torch.from_numpy(_records(scorer.score_batch(sh, k), k))

    def gather_fn(t, dst):
        calls["gather"] += 1
        return shard.dist_gather(t, dst)

    rec, ranges = shard.score_sharded(score_fn, batch, rank, world, gather_fn)
    assert calls["gather"] == 1                       # exactly one collective on the path
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
        np.save(out_path, rec.numpy())
    else:
        assert rec is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    from oracle import harness, orc
    from pyascore_amd import synth
    from pyascore_amd.device import unpack_summary
    out = str(tmp_path / "rec.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = unpack_summary(np.load(out), 4)
    batch, settings = synth.make_batch("cfg3", n_psm=240, seed=77)
    want = harness.make_scorer(orc.OracleAscore, settings, kind="oracle").score_batch(batch, 4)
    for key in ("best_score", "n_sig", "best_sig", "ascores", "alt_mask"):
        assert np.array_equal(got[key], want[key]), key
