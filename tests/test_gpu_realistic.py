"""Spectra that cluster the way an instrument's do (r05 verdict, item 5): isotope envelopes, doublets 0.2-1.5 tolerances
apart, repeated m/z, integer-count intensities, 800-3000 peaks; phospho + fixed oxidation + n-terminal acetyl; under the
general settings also "sty" and "ST" neutral losses, four ion types, fragment charges up to 4.  Uniform noise decides a
marked count node or a one-neighbour ion once in thousands of PSMs; here peaks sit next to each other and next to the
fragments by construction.  >= 2048 PSMs of either flavour through every route of the `path` fixture, against the
reference's own C++ core (run on all host cores: oracle/par_check.py); a sample PSM by PSM through score() with every
per-signature record compared."""
import numpy as np
import pytest

from conftest import checker_kind
from oracle import harness, orc, par_check
from pyascore_amd import synth
from test_gpu_parity import path  # noqa: F401  (the route fixture)

pytestmark = pytest.mark.gpu

N_PSM = 2048
_cache = {}


def _case(flavour):
    if flavour not in _cache:
        general = flavour == "general"
        batch, settings = synth.make_realistic(N_PSM, seed=6001 if general else 6002, general=general,
                                               max_sites=9 if general else 12, max_mod=3 if general else 5)
        k = int(batch["n_of_mod"].max())
        want = par_check.score_batch_parallel(settings, batch, k, kind=checker_kind())
        _cache[flavour] = (batch, settings, want, k)
    return _cache[flavour]


@pytest.mark.parametrize("flavour", ["general", "plain"])
def test_realistic_clusters_match_the_reference(flavour, path):
    from pyascore_amd import PyAscore
    batch, settings, want, k = _case(flavour)
    got = harness.make_scorer(PyAscore, settings).score_batch(batch)
    assert got["ascores"].shape[1] == k
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        assert bad.size == 0, "%s differs for PSMs %s" % (key, bad[:10])


@pytest.mark.parametrize("flavour", ["general", "plain"])
def test_realistic_clusters_record_by_record(flavour):
    """score() + pep_scores of every 32nd PSM: counts, scores and totals of every site assignment."""
    from pyascore_amd import PyAscore
    batch, settings, _, _ = _case(flavour)
    idx = np.arange(0, batch["n_psm"], 32)
    sub = synth.pack_batch([dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"],
                                 max_charge=kw["max_fragment_charge"], aux_pos=kw.get("aux_mod_pos", ()),
                                 aux_mass=kw.get("aux_mod_mass", ()))
                            for kw in (synth.unpack_psm(batch, int(i)) for i in idx)])
    got = harness.collect(harness.make_scorer(PyAscore, settings), sub, synth.unpack_psm)
    want = harness.collect(harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind()), sub, synth.unpack_psm)
    assert harness.compare(got, want, exact_float=True) == []


def test_the_generator_makes_what_it_says():
    """(cheap, no checker) peak counts, doublets within 1.5 tolerances, repeated m/z and tied intensities are there."""
    batch, settings = synth.make_realistic(64, seed=1, general=True)
    n = np.diff(batch["peak_off"])
    assert n.min() >= 700 and n.max() <= 3300
    d = np.diff(batch["mz"])
    inside = np.ones(d.size, bool)
    inside[batch["peak_off"][1:-1] - 1] = False
    assert (d[inside] == 0).sum() > 64 and ((d[inside] > 0) & (d[inside] < 1.5 * settings["mz_error"])).sum() > 640
    assert np.all(batch["intensity"] == np.floor(batch["intensity"])) and np.unique(batch["intensity"]).size < batch["intensity"].size // 20
