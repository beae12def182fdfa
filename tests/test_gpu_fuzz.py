"""Randomised parity: random scorer settings (mod groups incl. termini, ion types, neutral-loss
groups, tolerances) x random small batches (lengths, site counts, charges, fixed mods incl. the
n-terminus, sorted and unsorted peaks), HIP path vs the CPU checker on this box, bit for bit."""
import os

import numpy as np
import pytest

from conftest import checker_kind

from oracle import harness, orc
from pyascore_amd import synth

pytestmark = pytest.mark.gpu

from fuzzcase import random_case as _random_case


# PYA_FUZZ_SEEDS=a:b widens the sweep for a one-off soak (default: 40 seeds)
_LO, _HI = (int(x) for x in os.environ.get("PYA_FUZZ_SEEDS", "0:40").split(":"))


# seeds a soak found something with (each is a regression test now):
#   270 -- a batch for the single-launch kernel whose longest peptide has few site assignments (its caps
#          were accounted with the fused kernel's PSMs and left out of that launch's sizing)
_FOUND = [270]


@pytest.mark.parametrize("seed", list(range(_LO, _HI)) + [s for s in _FOUND if not _LO <= s < _HI])
def test_random_settings_and_batches(seed, monkeypatch):
    rng = np.random.default_rng(9000 + seed)
    # the routes a batch can take: single launch (small batches, default), or the kernel-per-stage
    # path with the fused score + localize kernel and the lean localize instantiation / with them
    # declining everything / general only / with the sort emulation forced / without the fused kernel
    route = seed % 6
    monkeypatch.setenv("PYA_PLAIN_MIN", "0")
    if route != 0:
        monkeypatch.setenv("PYA_NO_TINY", "1")
    if route == 5:
        monkeypatch.setenv("PYA_NO_FUSED", "1")
    if route == 2:
        monkeypatch.setenv("PYA_DEBUG", "512")
    elif route == 3:
        monkeypatch.setenv("PYA_NO_PLAIN", "1")
    elif route == 4:
        monkeypatch.setenv("PYA_DEBUG", "1024")
    settings, batch = _random_case(rng)
    if batch["n_psm"] == 0:
        pytest.skip("empty draw")
    from pyascore_amd import PyAscore
    gpu = harness.make_scorer(PyAscore, settings)
    kind = checker_kind()
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=kind)
    got = gpu.score_batch(batch)
    want = chk.score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        assert bad.size == 0, "%s differs for PSMs %s (settings %s)" % (key, bad[:5], settings)
    # and the full per-PSM API on a few of them
    sub = synth.slice_batch(batch, 0, min(4, batch["n_psm"]))
    a = harness.collect(gpu, sub, synth.unpack_psm)
    b = harness.collect(chk, sub, synth.unpack_psm)
    assert harness.compare(a, b, exact_float=True) == []
