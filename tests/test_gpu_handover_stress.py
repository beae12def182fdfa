"""The one-PSM hand-over under load (r04: one suite run returned empty pep_scores for a score() PSM; r05: this test
caught it again -- best_sequence '' at the first score() of a PSM that made the one-PSM workspace grow -- and
scripts/handover_repro.py found the cause: the workspace was zeroed by a null-stream hipMemset that could still be running
while the kernel, on a non-blocking stream, had already started; profiles/r05_handover_race.md).

PyAscore.score() hands one PSM to a single-launch kernel through a pinned block and polls a sequence word
(csrc/host_one.cpp); pep_scores replays the PSM lazily (pya_rescore_last_keep); score_batch of one PSM reuses the same
staging.  This test drives those three entry points in a seeded random interleaving for >= 50 000 calls over mixed shapes
(cfg1 / cfg2 / cfg3 shapes, fragment charge 1 and 2, a few heavier PSMs that take the plan path), every result compared
with the reference's C++ core (answers precomputed per distinct PSM), while a SECOND PROCESS keeps the same GPU busy
with batches of its own.  PYA_STRESS_CALLS overrides the call count (soaks)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, checker_kind
from oracle import harness, orc
from pyascore_amd import synth

pytestmark = pytest.mark.gpu

_LOAD = r"""
import os, sys, time
sys.path.insert(0, %r)
from pyascore_amd import PyAscore, synth
batch, st = synth.make_batch("cfg2", n_psm=20000, seed=5)
s = PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], st["mz_error"], st["fragment_types"])
print("ready", flush=True)
n = 0
while not os.path.exists(sys.argv[1]):
    s.score_batch(batch)
    n += 1
print("batches", n, flush=True)
"""


def _pool():
    psms = []
    for cfg, n, seed, over in (("cfg1", 40, 1, {}), ("cfg2", 200, 2, {}), ("cfg3", 300, 3, {}), ("cfg2", 60, 4, dict(max_charge=2)),
                               ("cfg3", 40, 5, dict(max_charge=2))):
        batch, settings = synth.make_batch(cfg, n_psm=n, seed=9000 + seed, **over)
        for i in range(n):
            psms.append(synth.unpack_psm(batch, i))
    # a few PSMs with thousands of site assignments: too big for the one-PSM kernel's LDS, they take the plan path
    batch, _ = synth.make_batch("cfg5", n_psm=4, seed=9100)
    for i in range(4):
        psms.append(synth.unpack_psm(batch, i))
    return psms, dict(settings, mz_error=0.05)


def test_fifty_thousand_interleaved_calls_with_a_busy_neighbour(tmp_path):
    from pyascore_amd import PyAscore
    n_calls = int(os.environ.get("PYA_STRESS_CALLS", "50000"))
    psms, settings = _pool()
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind())
    want = []
    for kw in psms:
        chk.score(**kw)
        raw = chk.raw_pep_scores()                           # (the records as arrays: no Python dict per record)
        want.append(dict(seq=chk.best_sequence, score=np.float32(chk.best_score), asc=np.asarray(chk.ascores, np.float32).copy(),
                         alt=[np.asarray(a).copy() for a in chk.alt_sites], n=int(raw["weighted_score"].shape[0]),
                         ws=np.asarray(raw["weighted_score"], np.float32), counts=np.asarray(raw["counts"], np.int32)))
    stop = str(tmp_path / "stop")
    load = subprocess.Popen([sys.executable, "-c", _LOAD % ROOT, stop], stdout=subprocess.PIPE, text=True)
    try:
        assert load.stdout.readline().strip() == "ready"
        gpu = harness.make_scorer(PyAscore, settings)
        rng = np.random.default_rng(20261004)
        calls = dict(score=0, records=0, batch_of_one=0, late_records=0)
        pending = None                                       # a PSM whose records are read only after other work
        done = 0
        t0 = time.time()
        while done < n_calls:
            i = int(rng.integers(len(psms)))
            kw, w = psms[i], want[i]
            what = rng.random()
            if what < 0.12:
                # score_batch of ONE PSM (reuses the one-PSM staging) -- first the records of a pending PSM, if any
                one = synth.pack_batch([dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"],
                                             max_charge=kw["max_fragment_charge"])])
                got = gpu.score_batch(one)
                assert np.float32(got["best_score"][0]) == w["score"] and int(got["n_sig"][0]) == w["n"], (done, i)
                assert np.array_equal(got["ascores"][0][: w["asc"].size], w["asc"]), (done, i)
                calls["batch_of_one"] += 1
                done += 1
                continue
            gpu.score(**kw)
            calls["score"] += 1
            done += 1
            assert gpu.best_sequence == w["seq"] and np.float32(gpu.best_score) == w["score"], (done, i)
            assert np.array_equal(gpu.ascores, w["asc"]), (done, i)
            if what < 0.45 and w["n"] > 600:
                gpu._ensure_kept()                           # thousands of records: the bulk arrays, not a dict per record
                got = gpu.batch_pep_scores()
                assert got["weighted_score"].shape[0] == w["n"], (done, i)
                assert np.array_equal(got["weighted_score"], w["ws"]) and np.array_equal(got["counts"], w["counts"]), (done, i)
                calls["records"] += 1
                done += 1
            elif what < 0.45:
                ps = gpu.pep_scores                          # lazy replay of the PSM just scored
                assert len(ps) == w["n"], "empty / short pep_scores after %d calls (PSM %d): %d of %d" % (done, i, len(ps), w["n"])
                assert np.array_equal(np.asarray([p["weighted_score"] for p in ps], np.float32), w["ws"]), (done, i)
                assert np.array_equal(np.asarray([p["counts"] for p in ps], np.int32), w["counts"]), (done, i)
                calls["records"] += 1
                done += 1
            elif what < 0.55:
                alt = gpu.alt_sites
                assert len(alt) == len(w["alt"]) and all(np.array_equal(a, b) for a, b in zip(alt, w["alt"])), (done, i)
                # ... and the records once more, after the properties were read
                if w["n"] <= 600:
                    assert len(gpu.pep_scores) == w["n"], (done, i)
                calls["late_records"] += 1
                done += 1
        took = time.time() - t0
    finally:
        open(stop, "w").close()
        out = load.communicate(timeout=300)[0]
    assert "batches" in out and int(out.split()[-1]) >= 1, out     # the neighbour really ran beside it
    print("%d calls in %.1f s beside %s of a second process: %s" % (done, took, out.strip(), calls))
    assert calls["score"] > 0.6 * n_calls * 0.5 and calls["records"] > 0 and calls["batch_of_one"] > 0


def test_the_first_call_after_the_workspace_grows():
    """The race behind the empty results, with its window made wide (debug switch PYA_SLOW_NULL_STREAM: 256 MB of
    null-stream memset queued in front of whatever the null stream does next).  Before the fix 9 685 of 9 813 such first
    calls came back with n_sig 0 / best_score -1; the workspace is now zeroed on the kernel's own stream and waited for."""
    from pyascore_amd import PyAscore
    psms, settings = _pool()
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind())
    picks = [0, 50, 300, 640, 641, 642, 643]
    want = {}
    for i in picks:
        chk.score(**psms[i])
        want[i] = (chk.best_sequence, np.float32(chk.best_score))
    for trial in range(60):
        gpu = harness.make_scorer(PyAscore, settings)
        gpu.set_debug("PYA_SLOW_NULL_STREAM", "1")
        # the first call allocates the workspace, a PSM with thousands of site assignments makes it grow
        for i in (picks[trial % 3], picks[3 + trial % 4], picks[(trial + 1) % 3]):
            gpu.score(**psms[i])
            assert (gpu.best_sequence, np.float32(gpu.best_score)) == want[i], (trial, i)
