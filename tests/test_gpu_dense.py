"""Dense spectra (r05 verdict, items 4 and 6): peak classes above 640 peaks are binned by SELECTION (bin_select.hip.h:
per-window histograms over the top bits of the intensity keys, a threshold bucket per window, then the all-pairs ranking
over the survivors only) instead of ranking every peak against every window mate.  The retained table -- hence every
result -- must not depend on the route: selection, the all-pairs kernel (PYA_BIN_SELECT_MIN huge), the serial
std::nth_element emulation (PYA_DEBUG=128) and the reference's own C++ core agree on

  * 1 500- and 4 000-peak spectra with isotope satellites (bench.py's dense legs), continuous and count-like intensities;
  * every hand-over case of the key format (test_gpu_parity.test_binning_keys_and_their_hand_overs) with the selection
    route forced on the sparse spectra too (PYA_BIN_SELECT_MIN=0);
  * survivors that do not fit their slots (PYA_BIN_SELECT_SCAP=64: nearly every spectrum is handed over);
  * flat windows (everything in one bucket), one window for the whole spectrum, more than 64 windows."""
import numpy as np
import pytest

import switches
from conftest import checker_kind
from oracle import harness, orc, par_check
from pyascore_amd import synth

pytestmark = pytest.mark.gpu

KEYS = ("n_sig", "best_sig", "best_score", "alt_mask", "ascores")
ROUTES = {"select": {}, "all_pairs": {"PYA_BIN_SELECT_MIN": "1000000"}, "exact": {"PYA_DEBUG": "128"},
          "select_forced": {"PYA_BIN_SELECT_MIN": "0"}, "select_overflow": {"PYA_BIN_SELECT_MIN": "0", "PYA_BIN_SELECT_SCAP": "64"}}


def _run(gpu, monkeypatch, route, batch):
    for name in ("PYA_BIN_SELECT_MIN", "PYA_BIN_SELECT_SCAP", "PYA_DEBUG"):
        monkeypatch.delenv(name, raising=False)
    for name, v in ROUTES[route].items():
        monkeypatch.setenv(name, v)
    switches.from_env(gpu)
    out = gpu.score_batch(batch)
    for name in ROUTES[route]:
        monkeypatch.delenv(name, raising=False)
    switches.from_env(gpu)
    return out


def _same(got, want, what):
    for key in KEYS:
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        assert bad.size == 0, "%s: %s differs for PSMs %s" % (what, key, bad[:10])


def _gpu(settings):
    from pyascore_amd import PyAscore
    return harness.make_scorer(PyAscore, settings)


@pytest.mark.parametrize("n_noise,n_psm", [(1500, 384), (4000, 256), (7600, 64)])
def test_dense_spectra_on_every_binning_route(n_noise, n_psm, monkeypatch):
    desc = synth.describe("cfg2", n_psm=n_psm, seed=7000 + n_noise, n_noise=n_noise, isotopes=True)
    batch = synth.make_slice(desc)
    settings = desc["settings"]
    assert np.diff(batch["peak_off"]).min() > 640
    k = int(batch["n_of_mod"].max())
    gpu = _gpu(settings)
    it = batch["intensity"]
    cases = {"continuous": it, "counts": np.floor(it / np.median(it) * 40.0) + 1.0,
             "narrow": 1000.0 + (it % 1.0) * 300.0}                  # (everything within a third of an octave: few buckets)
    for name, inten in cases.items():
        b2 = dict(batch, intensity=np.ascontiguousarray(inten))
        want = par_check.score_batch_parallel(settings, b2, k, kind=checker_kind())
        for route in ("select", "all_pairs", "exact", "select_overflow"):
            _same(_run(gpu, monkeypatch, route, b2), want, "%s / %s" % (name, route))


def test_selection_on_the_hand_over_cases_of_the_keys(monkeypatch):
    """The regimes of test_binning_keys_and_their_hand_overs on cfg2's sparse spectra, selection forced."""
    batch, settings = synth.make_batch("cfg2", n_psm=300, seed=4242)
    it = batch["intensity"]
    rng = np.random.default_rng(9)
    ulp = np.spacing(it)
    idx = rng.permutation(it.size)
    half = it.size // 2
    near = it.copy()
    near[idx[:half]] = np.floor(it[idx[:half]] / 64.0) * 64.0 + 1.0
    near[idx[:half]] += ulp[idx[:half]] * rng.integers(0, 4, half)
    top = it.copy()
    for i in range(batch["n_psm"]):
        a, b = batch["peak_off"][i], batch["peak_off"][i + 1]
        order = a + np.argsort(it[a:b])[::-1][:40]
        top[order] = 50000.0 + np.spacing(50000.0) * rng.integers(0, 6, order.size)
    wide = it.copy()
    wide[idx[: it.size // 5]] = 0.0
    wide[idx[it.size // 5: it.size // 4]] = 5e-324
    wide[idx[it.size // 4: it.size // 3]] *= 1e-30
    wide[idx[it.size // 3: it.size // 2]] *= 1e30
    neg = it.copy()
    neg[idx[: it.size // 10]] *= -1.0
    neg[idx[it.size // 10: it.size // 8]] = -0.0
    cases = {"plain": it, "near": near, "top": top, "wide": wide, "negative": neg,
             "coarse": np.floor(it / np.median(it) * 3.0) + 1.0, "counts": np.floor(it / np.median(it) * 40.0) + 1.0,
             "flat": np.ones_like(it)}
    gpu = _gpu(settings)
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind())
    for name, inten in cases.items():
        b2 = dict(batch, intensity=inten)
        got = _run(gpu, monkeypatch, "select_forced", b2)
        want = chk.score_batch(b2, got["ascores"].shape[1])
        _same(got, want, name)
        _same(_run(gpu, monkeypatch, "select_overflow", b2), want, name + " / overflow")
    # peaks out of m/z order: declined by the order check
    mz, inten = batch["mz"].copy(), it.copy()
    for i in range(batch["n_psm"]):
        a, b = batch["peak_off"][i], batch["peak_off"][i + 1]
        p = rng.permutation(b - a)
        mz[a:b], inten[a:b] = mz[a:b][p], inten[a:b][p]
    b2 = dict(batch, mz=mz, intensity=inten)
    got = _run(gpu, monkeypatch, "select_forced", b2)
    _same(got, chk.score_batch(b2, got["ascores"].shape[1]), "shuffled")


@pytest.mark.parametrize("bin_size", [100.0, 25.0, 2000.0])
def test_selection_with_other_window_widths(bin_size, monkeypatch):
    """bin_size 25 makes 76 windows (more than the keys have room for: handed over), 2 000 one window for everything."""
    desc = synth.describe("cfg2", n_psm=96, seed=7100, n_noise=1200, isotopes=True)
    batch = synth.make_slice(desc)
    settings = dict(desc["settings"], bin_size=bin_size)
    gpu = _gpu(settings)
    want = par_check.score_batch_parallel(settings, batch, int(batch["n_of_mod"].max()), kind=checker_kind())
    for route in ("select", "all_pairs", "select_overflow"):
        _same(_run(gpu, monkeypatch, route, batch), want, "bin_size %g / %s" % (bin_size, route))
