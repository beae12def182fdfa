"""A batch of mixed shapes runs its two independent kernel chains side by side (r06: pya_plan::fork -- the fused score +
localize kernels on the handle's side stream, the scoring + localize kernels of the other PSMs on the caller's, forked
after the binning and joined at the end of the run).  Forked or not (PYA_NO_FORK) the results are the reference's; runs
enqueued back to back on one stream, and on two streams at once, do not disturb each other; the timing ring reports the
fused family from the side stream's own events."""
import numpy as np
import pytest
import torch

import switches
from conftest import checker_kind
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan, unpack_summary

pytestmark = pytest.mark.gpu

KEYS = ("n_sig", "best_sig", "best_score", "alt_mask", "ascores")


def _same(got, want, what):
    for key in KEYS:
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        assert bad.size == 0, "%s: %s differs for PSMs %s" % (what, key, bad[:10])


def test_forked_and_unforked_runs_agree_with_the_reference(monkeypatch):
    batch, settings = synth.make_batch("cfg3", n_psm=6000, seed=6100)          # fused PSMs and count-node PSMs in one batch
    gpu = harness.make_scorer(PyAscore, settings)
    got = gpu.score_batch(batch)
    want = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind()).score_batch(batch, got["ascores"].shape[1])
    _same(got, want, "forked")
    monkeypatch.setenv("PYA_NO_FORK", "1")
    switches.from_env(gpu)
    _same(gpu.score_batch(batch), want, "PYA_NO_FORK")
    monkeypatch.delenv("PYA_NO_FORK")
    switches.from_env(gpu)


def test_runs_back_to_back_and_on_two_streams():
    batch, settings = synth.make_batch("cfg3", n_psm=4000, seed=6101)
    gpu = harness.make_scorer(PyAscore, settings)
    want = gpu.score_batch(batch)
    dev = torch.device("cuda", 0)
    mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
    plans = [DevicePlan(gpu, batch, timing=True), DevicePlan(gpu, batch, timing=True)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rep in range(6):                                                       # both plans share the handle's side stream
        for pl, st in zip(plans, streams):
            with torch.cuda.stream(st):
                pl.run(mz, it)
    torch.cuda.synchronize()
    for pl in plans:
        pl.check()
        got = unpack_summary(pl.packed_summary().cpu().numpy(), pl.max_k)
        k = want["ascores"].shape[1]
        for key in KEYS:
            assert np.array_equal(np.asarray(got[key]).reshape(np.asarray(want[key]).shape) if key != "ascores" else got[key][:, :k], want[key]), key
        ms, n = pl.timings_sum()
        assert n == 6 and ms[0] > 0 and ms[2] > 0 and ms[1] > 0 and ms[3] > 0   # binning, scoring, the fused family (side stream), localize
        pl.close()
