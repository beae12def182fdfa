"""GPU parity tests proper: the HIP path (through the C ABI / PyAscore) against the committed
golden vectors, the CPU restatement and -- where its prebuilt library travelled -- the
reference's own C++ core.  Integer results bit-exact; float scores bit-exact against a checker
running on the same host libm, rtol 1e-6 against the golden files."""
import os

import numpy as np
import pytest

import switches

from conftest import GOLDEN, golden_cases, checker_kind
from oracle import harness, orc
from pyascore_amd import synth

pytestmark = pytest.mark.gpu


def test_loaded_library_is_built_from_this_tree():
    """The library this suite runs names the SHA-256 of csrc/ + include/ it was linked from (csrc/version.cpp);
    recomputed from the tree on this box it must be the same, and every object's own record (flags + digest of
    the compiler's dependency list) must match too: a stale object cannot pass for HEAD."""
    from pyascore_amd import _lib, build
    lib = _lib.load()
    assert ("src=" + build.tree_digest()).encode() in lib.pya_version(), lib.pya_version()
    for name in sorted(os.listdir(build.CSRC)):
        if name.endswith((".hip", ".cpp")):
            obj = os.path.join(build.CSRC, name + ".o")
            deps = build._deps_of(obj)
            assert deps and open(obj + ".flags").read().split("\n")[1] == build._digest(deps), name


@pytest.fixture(params=["fused", "lean_localize", "general_localize", "lean_declines", "always_sort", "fused_always_sort",
                        "fused_replay", "general_serial_replay", "big_records", "general_lists", "hash_exact", "hash_declines",
                        "hash_small_lists", "hash_no_closed_form", "score_walkers", "score_few_nodes"])
def path(request, monkeypatch):
    """Every batch runs on every route a PSM can take (results stay exact on all of them):
    fused                  plain PSMs (no neutral losses, one ion type per direction) with few site assignments on the
                           fused score + localize kernel, the other plain ones on the lean localize instantiation (default)
    lean_localize          without the fused kernel (PYA_NO_FUSED=1)
    general_localize       every PSM on the general instantiation (PYA_NO_PLAIN=1): hash route for site-determining ions
    lean_declines          the fused kernel and the lean instantiation decline every PSM (PYA_DEBUG=512): hand-over lists
    always_sort            the std::sort emulation runs even for a unique best PepScore (PYA_DEBUG=1024), without / with
    fused_always_sort      the fused kernel
    fused_replay           the fused kernel replays every (competitor, direction) task serially (PYA_DEBUG=2048)
    general_serial_replay  the list-based general instantiation with whole-task serial replays (PYA_DEBUG=4096)
    general_lists          the list-based general route the hash route replaced (PYA_NO_LOC_HASH=1)
    hash_exact             the hash route sends every in-span ion through the exact run walk (PYA_DEBUG=16384)
    hash_no_closed_form    ... sends an ion with one neighbour there too (PYA_DEBUG=0x20000000); by default an ion next to one
                           ion of the winner outside the span (+ its twin), or next to one other ion of the span that is as
                           alone, is decided in closed form
    hash_declines          the hash route declines every PSM (PYA_DEBUG=8192: hand-over list, list-based kernel)
    hash_small_lists       room for the pair lists of short spans only (PYA_HASH_PP=2): a share of the PSMs is handed over
    score_walkers          score_signatures under general settings with one walker per signature (PYA_NO_NODES=1)
    score_few_nodes        ... with room for 150 shared nodes per direction: some directions fall back to the walkers
    big_records            PSMs with thousands of site assignments on the older route (count records, sort in localize)
    (Routes measured slower and deleted in round 4 -- several PSMs per wavefront, binning inside the fused kernel, the
    hash / recount launches with the retained table staged in LDS -- are described in DESIGN.md section 10.)"""
    for v in ("PYA_NO_PLAIN", "PYA_NO_LOC_HASH", "PYA_HASH_PP", "PYA_NO_NODES", "PYA_NODE_CAP", "PYA_NO_FUSED", "PYA_DEBUG",
              "PYA_NO_BIG_INLINE"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("PYA_PLAIN_MIN", "0")       # batches under 512 PSMs skip the lean kernels by default
    monkeypatch.setenv("PYA_NO_TINY", "1")         # ... and those of up to 64 the three kernels altogether
    env = {
        "fused": {},
        "lean_localize": {"PYA_NO_FUSED": "1"},
        "general_localize": {"PYA_NO_PLAIN": "1"},
        "lean_declines": {"PYA_DEBUG": "512"},
        "always_sort": {"PYA_NO_FUSED": "1", "PYA_DEBUG": "1024"},
        "fused_always_sort": {"PYA_DEBUG": "1024"},
        "fused_replay": {"PYA_DEBUG": "2048"},
        "general_serial_replay": {"PYA_NO_PLAIN": "1", "PYA_NO_LOC_HASH": "1", "PYA_DEBUG": "4096"},
        "general_lists": {"PYA_NO_PLAIN": "1", "PYA_NO_LOC_HASH": "1"},
        "hash_exact": {"PYA_NO_PLAIN": "1", "PYA_DEBUG": "16384"},
        "hash_no_closed_form": {"PYA_NO_PLAIN": "1", "PYA_DEBUG": str(0x20000000)},
        "hash_declines": {"PYA_NO_PLAIN": "1", "PYA_DEBUG": "8192"},
        "hash_small_lists": {"PYA_NO_PLAIN": "1", "PYA_HASH_PP": "2"},
        "score_walkers": {"PYA_NO_NODES": "1"},
        "score_few_nodes": {"PYA_NODE_CAP": "150"},
        "big_records": {"PYA_NO_BIG_INLINE": "1"},
    }[request.param]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return request.param


def _gpu(settings):
    from pyascore_amd import PyAscore
    return harness.make_scorer(PyAscore, settings)


def _checker(settings):
    kind = checker_kind()
    return harness.make_scorer(orc.OracleAscore, settings, kind=kind)


@pytest.mark.parametrize("route", ["single_launch", "three_kernels"])
@pytest.mark.parametrize("case", golden_cases())
def test_public_api_matches_golden(case, route, monkeypatch):
    """score() + every property of PyAscore, PSM by PSM, against the reference's outputs -- on the
    fused single-launch kernel batches of a few PSMs take by default, and on the three-kernel path
    (PYA_NO_TINY=1)."""
    if route == "three_kernels":
        monkeypatch.setenv("PYA_NO_TINY", "1")
    else:
        monkeypatch.delenv("PYA_NO_TINY", raising=False)
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    got = harness.collect(_gpu(settings), batch, synth.unpack_psm)
    assert harness.compare(got, expected, exact_float=False, rtol=1e-6, atol=1e-6) == []
    # same host libm as the generator of the goldens -> expect bit equality too
    assert harness.compare(got, expected, exact_float=True) == []


@pytest.mark.parametrize("case", golden_cases())
def test_batch_matches_golden(case, path):
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    got = _gpu(settings).score_batch(batch)
    k = got["ascores"].shape[1]
    for key in ("best_sig", "n_sig", "alt_mask"):
        assert np.array_equal(got[key], expected[key][..., :k] if expected[key].ndim == 2 else expected[key]), key
    assert np.array_equal(got["best_score"], expected["best_score"])
    assert np.array_equal(got["ascores"], expected["ascores"][:, :k])


@pytest.mark.parametrize("cfg,n,seed,override", [
    ("cfg1", 2000, 201, {}),
    ("cfg2", 4000, 202, {}),
    ("cfg3", 4000, 203, {}),
    ("cfg4", 48, 204, {}),
    ("cfg5", 24, 205, {}),
    ("cfg2", 600, 206, dict(fragment_types="yb", max_charge=2)),
    ("cfg2", 200, 207, dict(fragment_types="Zc", max_charge=3, neutral_loss=("STY", 18.01528))),
    ("cfg2", 600, 208, dict(mz_error=0.5)),
    ("cfg3", 1500, 209, dict(mz_error=0.3, max_charge=2)),
    ("cfg4", 32, 211, dict(mz_error=0.002)),      # a tolerance below the rounding of the sums: the hash route declines, lists take over
    ("cfg4", 32, 212, dict(mz_error=0.2)),        # a wide one: many ions with neighbours, the exact run walk
])
def test_batch_matches_checker(cfg, n, seed, override, path):
    """Fresh seeded batches, HIP vs the CPU checker on this box: bit-exact everywhere."""
    batch, settings = synth.make_batch(cfg, n_psm=n, seed=seed, **override)
    got = _gpu(settings).score_batch(batch)
    want = _checker(settings).score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        assert bad.size == 0, "%s differs for PSMs %s" % (key, bad[:10])


def test_unsorted_spectrum_and_ties(path):
    """Peaks given in arbitrary order (the API does not require sorted m/z)."""
    batch, settings = synth.make_batch("cfg3", n_psm=300, seed=210)
    rng = np.random.default_rng(1)
    mz, it = batch["mz"].copy(), batch["intensity"].copy()
    for i in range(batch["n_psm"]):
        a, b = batch["peak_off"][i], batch["peak_off"][i + 1]
        p = rng.permutation(b - a)
        mz[a:b], it[a:b] = mz[a:b][p], it[a:b][p]
    shuffled = dict(batch, mz=mz, intensity=it)
    got = _gpu(settings).score_batch(shuffled)
    want = _checker(settings).score_batch(shuffled, got["ascores"].shape[1])
    for key in want:
        assert np.array_equal(got[key], want[key]), key


def test_equal_intensities_follow_nth_element(monkeypatch):
    """Which of several equally intense peaks of a window are retained, and their rank order, is
    whatever std::nth_element + std::sort leave in the reference (Spectra.cpp:24-41).  bin_spectra
    ranks with a strict-compare sweep, notices equal intensities through a count identity and then
    emulates the two library calls window by window: results must equal the reference's on spectra
    full of ties, and forcing that route for every spectrum (PYA_DEBUG=128) must change nothing."""
    batch, settings = synth.make_batch("cfg2", n_psm=400, seed=77)
    it = batch["intensity"]
    cases = {
        "coarse": np.floor(it / np.median(it) * 3.0) + 1.0,           # a handful of levels: ties everywhere
        "counts": np.floor(it / np.median(it) * 40.0) + 1.0,          # count-like: ties among weak peaks
        "flat": np.ones_like(it),                                     # every peak ties with every other
    }
    gpu = _gpu(settings)
    chk = _checker(settings)
    for name, inten in cases.items():
        tied = dict(batch, intensity=inten)
        monkeypatch.delenv("PYA_DEBUG", raising=False)
        switches.from_env(gpu)                                  # (the switches are read once per scorer)
        got = gpu.score_batch(tied)
        want = chk.score_batch(tied, got["ascores"].shape[1])
        for key in want:
            assert np.array_equal(got[key], want[key]), (name, key)
        monkeypatch.setenv("PYA_DEBUG", "128")
        switches.from_env(gpu)
        forced = gpu.score_batch(tied)
        monkeypatch.delenv("PYA_DEBUG", raising=False)
        switches.from_env(gpu)
        for key in got:
            assert np.array_equal(got[key], forced[key]), (name, key)
    # peaks out of m/z order AND equal intensities: the windows' input order decides
    rng = np.random.default_rng(11)
    mz, inten = batch["mz"].copy(), cases["counts"].copy()
    for i in range(batch["n_psm"]):
        a, b = batch["peak_off"][i], batch["peak_off"][i + 1]
        p = rng.permutation(b - a)
        mz[a:b], inten[a:b] = mz[a:b][p], inten[a:b][p]
    shuffled = dict(batch, mz=mz, intensity=inten)
    got = gpu.score_batch(shuffled)
    want = chk.score_batch(shuffled, got["ascores"].shape[1])
    for key in want:
        assert np.array_equal(got[key], want[key]), ("shuffled", key)
    # no ties at all: the forced route must still agree with the fast one
    monkeypatch.setenv("PYA_DEBUG", "128")
    switches.from_env(gpu)
    forced = gpu.score_batch(batch)
    monkeypatch.delenv("PYA_DEBUG", raising=False)
    switches.from_env(gpu)
    plain = gpu.score_batch(batch)
    for key in plain:
        assert np.array_equal(plain[key], forced[key]), key


def test_binning_keys_and_their_hand_overs(monkeypatch):
    """The common-case binning ranks with 31-bit composite keys (window | high word of the intensity, 20 mantissa
    bits, 2^32 of dynamic range below the spectrum's maximum: bin_core.hip.h, bin_fast) and hands over what they cannot
    decide.  Every regime against the checker: intensities that differ only below the key's precision, at the top of
    their windows and among the weak peaks; zero, denormal and huge intensities in one spectrum; negative and -0.0
    intensities; more than 64 windows (small bin_size); a whole spectrum inside one window; spectra of more than one
    384-peak block; and the forced general route (PYA_DEBUG=128) agreeing with the default on all of them."""
    batch, settings = synth.make_batch("cfg2", n_psm=300, seed=4242)
    it = batch["intensity"]
    rng = np.random.default_rng(9)
    ulp = np.spacing(it)
    near = it.copy()                                                   # pairs a few ulps apart: equal keys, different doubles
    idx = rng.permutation(it.size)
    half = it.size // 2
    near[idx[:half]] = np.floor(it[idx[:half]] / 64.0) * 64.0 + 1.0
    near[idx[:half]] += ulp[idx[:half]] * rng.integers(0, 4, half)
    top = it.copy()                                                    # the most intense peaks of every spectrum a few ulps apart
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
    cases = {"near": near, "top": top, "wide": wide, "negative": neg}
    gpu, chk = _gpu(settings), _checker(settings)

    def both(b2, name, g=gpu, c=chk):
        monkeypatch.delenv("PYA_DEBUG", raising=False)
        switches.from_env(g)
        got = g.score_batch(b2)
        want = c.score_batch(b2, got["ascores"].shape[1])
        for key in want:
            assert np.array_equal(got[key], want[key]), (name, key)
        monkeypatch.setenv("PYA_DEBUG", "128")
        switches.from_env(g)
        forced = g.score_batch(b2)
        monkeypatch.delenv("PYA_DEBUG", raising=False)
        switches.from_env(g)
        for key in got:
            assert np.array_equal(got[key], forced[key]), (name, "forced", key)
        # the same spectra one PSM per call: PyAscore.score() bins with the four wavefronts of its own kernel
        # (bin_core.hip.h: bin_fast_mw), which has its own sweeps, hand-overs and second sweep for equal keys
        few = synth.slice_batch(b2, 0, min(6, int(b2["n_psm"])))
        diff = harness.compare(harness.collect(g, few, synth.unpack_psm), harness.collect(c, few, synth.unpack_psm), exact_float=True)
        assert not diff, (name, "score()", diff)

    for name, inten in cases.items():
        both(dict(batch, intensity=inten), name)
    # a whole spectrum inside one 100 m/z window (one run of ~330 peaks), and spectra of several blocks
    squeezed = dict(batch, mz=400.0 + (batch["mz"] - 100.0) / 20.0)
    both(squeezed, "one window")
    # spectra of several 384-peak blocks: the peaks of five spectra merged under the first one's peptide
    po = batch["peak_off"]
    mzs, its, offs = [], [], [0]
    for j in range(0, 200, 5):
        m, t = batch["mz"][po[j]:po[j + 5]], it[po[j]:po[j + 5]]
        o = np.argsort(m, kind="stable")
        mzs.append(m[o]); its.append(t[o]); offs.append(offs[-1] + m.size)
    pick = np.arange(0, 200, 5)
    L = int(batch["pep_off"][1] - batch["pep_off"][0])
    big = dict(batch, n_psm=pick.size, mz=np.concatenate(mzs), intensity=np.concatenate(its),
               peak_off=np.asarray(offs, np.int64), pep=batch["pep"].reshape(-1, L)[pick].ravel().copy(),
               pep_off=np.arange(pick.size + 1, dtype=np.int64) * L, n_of_mod=batch["n_of_mod"][pick].copy(),
               max_charge=batch["max_charge"][pick].copy(), aux_off=np.zeros(pick.size + 1, np.int64))
    both(big, "blocks")
    # more than 64 windows: bin_size 10
    fine = dict(settings, bin_size=10.0)
    both(batch, "bin_size 10", _gpu(fine), _checker(fine))


@pytest.mark.parametrize("case", ["synth_cfg3", "edge_nl", "velos_zprec", "ties_cfg2"])
def test_bulk_pep_scores_match_golden(case):
    """pya_get_pep_scores_range: every localisation of every PSM of a retained batch
    in one call, in the reference's sorted order -- against the reference's pep_scores."""
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    gpu = _gpu(settings)
    gpu.score_batch(batch, keep=True)
    got = gpu.batch_pep_scores()
    assert np.array_equal(got["rec_off"], expected["ps_off"])
    assert np.array_equal(got["sig_bits"], expected["ps_bits"])
    assert np.array_equal(got["counts"], expected["ps_counts"])
    assert np.array_equal(got["total_fragments"], expected["ps_nfrag"])
    assert np.array_equal(got["scores"], expected["ps_scores"])
    assert np.array_equal(got["weighted_score"], expected["ps_ws"])
    # a sub-range carries the same rows
    n = batch["n_psm"]
    lo, hi = n // 3, max(n // 3 + 1, 2 * n // 3)
    part = gpu.batch_pep_scores(lo, hi)
    a, b = int(expected["ps_off"][lo]), int(expected["ps_off"][hi])
    assert np.array_equal(part["rec_off"], expected["ps_off"][lo:hi + 1] - a)
    assert np.array_equal(part["sig_bits"], expected["ps_bits"][a:b])
    assert np.array_equal(part["weighted_score"], expected["ps_ws"][a:b])
    with pytest.raises(ValueError):
        gpu.batch_pep_scores(0, n + 1)
    gpu.score_batch(batch)                     # not retained any more
    with pytest.raises(RuntimeError):
        gpu.batch_pep_scores()


def test_largest_shapes_end_to_end(path):
    """C(16,8) = 12 870 and C(45,3) = 14 190 site assignments per PSM (the limit is 15 000): the
    buckets with prefix sharing, multi-chunk sorting and the largest LDS footprints."""
    for n_sites, n_mod, L in ((16, 8, 30), (45, 3, 50)):
        batch, settings = synth.make_batch("cfg5", n_psm=3, seed=500 + n_sites, L=L, n_sites=n_sites, n_mod=n_mod)
        got = _gpu(settings).score_batch(batch)
        want = _checker(settings).score_batch(batch, got["ascores"].shape[1])
        for key in want:
            assert np.array_equal(got[key], want[key]), (n_sites, n_mod, key)


def test_longest_peptide_and_most_sites(path):
    """64 residues with 63 modifiable ones (the limits), one and two modifications, all ion types,
    charge 2: every 64-bit mask is full."""
    rng = np.random.default_rng(64)
    from pyascore_amd.synth import pack_batch
    psms = []
    for pep, k in (("S" * 63 + "K", 1), ("ST" * 31 + "YK", 2), ("K" + "Y" * 63, 1)):
        mass = np.array([synth.RESIDUE_MASS[c] for c in pep])
        frag = np.concatenate([np.cumsum(mass)[:-1] + synth.PROTON, np.cumsum(mass[::-1])[:-1] + synth.WATER + synth.PROTON])
        mz = np.sort(np.concatenate([frag[rng.random(frag.size) < 0.5] + rng.uniform(-0.01, 0.01), rng.uniform(100, 6000, 400)]))
        psms.append(dict(mz=mz, intensity=rng.lognormal(5.0, 1.0, mz.size), peptide=pep, n_of_mod=k, max_charge=2,
                         aux_pos=np.array([64], np.uint32), aux_mass=np.array([42.010565], np.float32)))
    batch = pack_batch(psms)
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.05,
                    fragment_types="bycz", neutral_losses=[])
    got = _gpu(settings).score_batch(batch)
    want = _checker(settings).score_batch(batch, got["ascores"].shape[1])
    for key in want:
        assert np.array_equal(got[key], want[key]), key


def test_many_tied_competitors():
    """Shapes with many single-move competitors (k * (n_sites - k) up to 126 for C(n,k) <= 15 000):
    at noise level most of them tie for the best score of their site and all of them have to be
    carried into the Ascore stage."""
    for n_sites, n_mod, L in ((17, 6, 24), (45, 3, 50), (14, 7, 20)):
        batch, settings = synth.make_batch("cfg5", n_psm=12, seed=300 + n_sites, L=L, n_sites=n_sites, n_mod=n_mod)
        # weak evidence: keep the noise, drop most signal peaks, so that many assignments tie
        rng = np.random.default_rng(n_sites)
        keep = rng.random(batch["mz"].size) < 0.85
        off = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), batch["peak_off"][:-1]))])
        thin = dict(batch, mz=batch["mz"][keep], intensity=batch["intensity"][keep], peak_off=off.astype(np.int64))
        got = _gpu(settings).score_batch(thin)
        want = _checker(settings).score_batch(thin, got["ascores"].shape[1])
        for key in want:
            assert np.array_equal(got[key], want[key]), (n_sites, n_mod, key)


def test_negative_residue_mass_and_crowded_lists():
    """Fragment lists are ascending only while every residue mass is positive (localize then skips
    its order check): a fixed modification heavier than its residue, negative, must still take
    the sorting route.  A wide tolerance makes ions find several partners, which sends tasks
    through the serial replay of the reference's greedy walk after the optimistic pass."""
    batch, settings = synth.make_batch("cfg2", n_psm=300, seed=91)
    rng = np.random.default_rng(3)
    aux_pos, aux_mass, aux_off = [], [], [0]
    for i in range(batch["n_psm"]):
        L = int(batch["pep_off"][i + 1] - batch["pep_off"][i])
        pos = rng.choice(np.arange(1, L + 1), size=2, replace=False)
        aux_pos += [int(p) for p in pos]
        aux_mass += [-250.0, 15.994915]
        aux_off.append(len(aux_pos))
    neg = dict(batch, aux_pos=np.array(aux_pos, np.uint32), aux_mass=np.array(aux_mass, np.float32),
               aux_off=np.array(aux_off, np.int64))
    for mz_error in (settings["mz_error"], 4.0):
        st = dict(settings, mz_error=mz_error)
        got = _gpu(st).score_batch(neg)
        want = _checker(st).score_batch(neg, got["ascores"].shape[1])
        for key in want:
            assert np.array_equal(got[key], want[key]), (key, mz_error)


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 20, 33, 64, 65, 200, 495, 1000, 3003, 4097, 15000])
def test_sort_emulation_matches_std_sort(n):
    """The on-device emulation of libstdc++ std::sort vs the real thing, ties included."""
    from pyascore_amd import PyAscore, _lib
    import ctypes as C
    s = PyAscore(100.0, 10, "STY", 79.966331)
    rng = np.random.default_rng(n)
    cases = [
        rng.integers(0, 4, n).astype(np.float32),                  # heavy ties
        rng.random(n).astype(np.float32),                          # no ties
        np.zeros(n, np.float32),                                   # all equal
        np.arange(n, dtype=np.float32),                            # ascending (worst for "greater")
        np.arange(n, dtype=np.float32)[::-1].copy(),               # already sorted
        np.round(rng.lognormal(3, 1, n), 0).astype(np.float32),    # realistic score ties
    ]
    # median-of-3 killer: drives introsort into its heapsort fallback for larger n
    if n >= 64:
        k = n // 2
        killer = np.zeros(n, np.float32)
        for i in range(k):
            killer[2 * i if 2 * i < n else n - 1] = i if i % 2 == 0 else k + i
        cases.append(-killer)
    for keys in cases:
        perm = np.zeros(n, np.uint32)
        rc = s._lib.pya_debug_sort(s._h, keys.ctypes.data_as(C.c_void_p), n, perm.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert np.array_equal(perm, orc.std_sort(keys))


def test_calculate_ambiguity_and_structure():
    """The reference's own structural tests (test/test_ascore.py:63-163) on the Velos PSMs."""
    import re
    from math import comb
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, "velos_z1.npz"))
    s = _gpu(settings)
    chk = _checker(settings)
    for i in range(batch["n_psm"]):
        kw = synth.unpack_psm(batch, i)
        s.score(**kw)
        chk.score(**kw)
        for alt in s.alt_sites:
            assert alt.shape[0] == np.unique(alt).shape[0]
        sites = [ind + 1 for ind, m in enumerate(re.finditer("[A-Z][^A-Z]*", s.best_sequence)) if "[80]" in m.group()]
        assert np.intersect1d(sites, np.concatenate(s.alt_sites)).shape[0] == 0
        ps = s.pep_scores
        assert len(ps) == comb(len(ps[0]["signature"]), kw["n_of_mod"])
        assert np.all(np.diff([p["weighted_score"] for p in ps]) <= 0)
        assert s.calculate_ambiguity(ps[0], ps[0]) == 0.0
        if len(ps) > 1:
            a = s.calculate_ambiguity(ps[0], ps[1])
            assert np.any(np.isclose(a, s.ascores))
            cps = chk.pep_scores
            for j in (1, len(ps) // 2, len(ps) - 1):
                assert a == chk.calculate_ambiguity(cps[0], cps[1])
                assert s.calculate_ambiguity(ps[0], ps[j]) == chk.calculate_ambiguity(cps[0], cps[j])
                assert s.calculate_ambiguity(ps[j], ps[0]) == chk.calculate_ambiguity(cps[j], cps[0])


def test_error_behaviour():
    from pyascore_amd import PyAscore
    s = PyAscore(100.0, 10, "STY", 79.966331)
    assert s.best_sequence == "" and s.best_score == -1.0 and s.ascores.size == 0 and s.pep_scores == []
    mz = np.array([150.0, 300.5, 420.25]); it = np.array([1.0, 2.0, 3.0])
    with pytest.raises(ValueError):
        s.score(mz, it, "PEPTIXDE", 1)                      # unknown residue: reference aborts
    with pytest.raises(ValueError):
        s.score(np.zeros(0), np.zeros(0), "PEPTIDE", 1)     # empty spectrum: reference is UB
    with pytest.raises(ValueError):
        s.score(mz.astype(np.float32), it, "PEPTIDE", 1)    # dtype mismatch (Cython ValueError)
    with pytest.raises(ValueError):
        s.score(np.arange(6.0)[::2], it, "PEPTIDE", 1)      # not C-contiguous
    with pytest.raises(TypeError):
        s.score(None, it, "PEPTIDE", 1)
    with pytest.raises(ValueError):
        s.score(mz, it, "PEPTIDE", 1, 0)                    # max_fragment_charge 0
    with pytest.raises(ValueError):
        s.score(np.array([500.0]), np.array([1.0]), "PEPTIDE", 1)   # no m/z window at all
    with pytest.raises(ValueError):
        PyAscore(100.0, 5, "STY", 79.966331)                # n_top < 10: reference reads garbage
    with pytest.raises(ValueError):
        PyAscore(100.0, 10, "STY", 79.966331, fragment_types="bx")
    s.score(mz, it, "PEPTIDE", 1)                           # still usable afterwards
    assert s.best_sequence == "PEPT[80]IDE"


# ---------------------------------------------------------------------------------------------
# Full BASELINE sizes: size-independent properties (the checker cannot run 100k+ PSMs in seconds)
# ---------------------------------------------------------------------------------------------
def _same(a, b, keys=("n_sig", "best_sig", "best_score", "ascores", "alt_mask")):
    for k in keys:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cfg,n,n_sample", [
    ("cfg2", 100_000, 1500),
    ("cfg3", 1_000_000, 1500),       # the whole 8-GPU config on one GPU
    ("cfg4", 250_000, 1000),         # general localize route: neutral losses, four ion types, charge 4
    ("cfg5", 50_000, 120),           # 3003 site assignments per PSM: prefix-shared walker, multi-chunk sort
])
def test_full_size_properties(cfg, n, n_sample):
    """Every BASELINE config at its full size (the reference's test/test_ascore.py:10-61 likewise runs
    every fixture in every setting): (1) determinism, (2) a batch scores exactly like its two halves,
    (3) PSM order does not matter, (4) a random sample agrees bit for bit with the reference's C++
    core on this box."""
    desc = synth.describe(cfg, n_psm=n, seed=4242)
    batch, settings = synth.make_slice(desc), desc["settings"]
    gpu = _gpu(settings)
    full = gpu.score_batch(batch)
    _same(gpu.score_batch(batch), full)
    half = n // 2
    lo, hi = gpu.score_batch(synth.slice_batch(batch, 0, half)), gpu.score_batch(synth.slice_batch(batch, half, n))
    k = full["ascores"].shape[1]
    for key in ("n_sig", "best_sig", "best_score"):
        assert np.array_equal(np.concatenate([lo[key], hi[key]]), full[key]), key
    for key in ("ascores", "alt_mask"):
        kl, kh = lo[key].shape[1], hi[key].shape[1]
        assert np.array_equal(lo[key], full[key][:half, :kl]) and not full[key][:half, kl:].any(), key
        assert np.array_equal(hi[key], full[key][half:, :kh]) and not full[key][half:, kh:].any(), key
    del lo, hi
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(n, n_sample, replace=False))
    sub = synth.pack_batch([dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"],
                                 n_of_mod=kw["n_of_mod"], max_charge=kw["max_fragment_charge"])
                            for kw in (synth.unpack_psm(batch, int(i)) for i in pick[::-1])])   # reversed order
    got = gpu.score_batch(sub)
    want = _checker(settings).score_batch(sub, got["ascores"].shape[1])
    _same(got, want)
    kk = got["ascores"].shape[1]
    for key in ("n_sig", "best_sig", "best_score"):
        assert np.array_equal(got[key][::-1], full[key][pick]), key
    assert np.array_equal(got["ascores"][::-1][:, :kk], full["ascores"][pick][:, :kk])
    assert np.array_equal(got["alt_mask"][::-1][:, :kk], full["alt_mask"][pick][:, :kk])


def test_peak_order_invariance_at_scale():
    """Shuffling the peaks inside every spectrum (unsorted-input path of bin_spectra) must not
    change any result."""
    batch, settings = synth.make_batch("cfg2", n_psm=20_000, seed=99)
    rng = np.random.default_rng(3)
    mz, it = batch["mz"].copy(), batch["intensity"].copy()
    off = batch["peak_off"]
    for i in range(batch["n_psm"]):
        p = rng.permutation(off[i + 1] - off[i])
        mz[off[i]:off[i + 1]], it[off[i]:off[i + 1]] = mz[off[i]:off[i + 1]][p], it[off[i]:off[i + 1]][p]
    gpu = _gpu(settings)
    _same(gpu.score_batch(dict(batch, mz=mz, intensity=it)), gpu.score_batch(batch))


def test_device_plan_and_shard_path_on_one_gpu():
    """pyascore_amd.device.DevicePlan + shard.score_sharded with world_size 1: records gathered
    from the device-resident path equal the host API's results."""
    import torch
    from pyascore_amd import shard
    from pyascore_amd.device import DevicePlan, unpack_summary
    batch, settings = synth.make_batch("cfg3", n_psm=3000, seed=31)
    gpu = _gpu(settings)
    want = gpu.score_batch(batch)
    k = int(batch["n_of_mod"].max())

    def score_fn(sh, max_k):
        dev = torch.device("cuda", gpu.device)
        plan = DevicePlan(gpu, sh, max_k=max_k)
        plan.run(torch.from_numpy(sh["mz"]).to(dev), torch.from_numpy(sh["intensity"]).to(dev))
        plan.check()
        return plan.packed_summary()

    rec, ranges = shard.score_sharded(score_fn, batch, 0, 1, lambda t, dst: [t])
    got = unpack_summary(rec.cpu().numpy(), k)
    _same(got, want)


def test_largest_spectra_and_limits():
    """Spectra at the documented 8192-peak limit (more than 64 KB of LDS per wave in bin_spectra
    and score_signatures), one peak, and the limit errors."""
    from pyascore_amd import PyAscore
    rng = np.random.default_rng(17)
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.05,
                    fragment_types="by", neutral_losses=[])
    psms = []
    for P in (8192, 8000, 5000, 1, 2):
        mz = np.sort(rng.uniform(100.0, 3000.0, P))
        psms.append(dict(mz=mz, intensity=rng.lognormal(5, 1, P), peptide="ASTLGYKRSTYAGK", n_of_mod=2, max_charge=2))
    batch = synth.pack_batch(psms)
    got = _gpu(settings).score_batch(batch)
    want = _checker(settings).score_batch(batch, got["ascores"].shape[1])
    _same(got, want)
    s = PyAscore(100.0, 10, "STY", 79.966331)
    with pytest.raises(ValueError, match="65535"):
        s.score(np.sort(rng.uniform(100.0, 3000.0, 65536)), np.ones(65536), "ASTK", 1)
    with pytest.raises(ValueError, match="length"):
        s.score(np.array([100.5, 200.5]), np.ones(2), "A" * 512, 0)
    s.score(np.array([100.5, 200.5]), np.ones(2), "A" * 65, 0)          # (the general kernel: 65 to 511 residues)
    assert s.best_sequence == "A" * 65
    with pytest.raises(ValueError, match="site assignments"):
        s.score(np.array([100.5, 200.5]), np.ones(2), "STSTSTSTSTSTSTSTSTSTSTSTSTSTSTSTSTSTSTST", 12)


def test_rccl_gather_path_single_rank():
    """bench.py's N > 1 step = plan.run + ONE dist.gather of the packed records over RCCL.  With
    one GPU only a world of one can be formed, which still drives the same calls on device
    tensors (backend "nccl" is RCCL on ROCm)."""
    import socket
    import torch
    import torch.distributed as dist
    from pyascore_amd import shard
    from pyascore_amd.device import DevicePlan, unpack_summary
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        batch, settings = synth.make_batch("cfg2", n_psm=5000, seed=55)
        gpu = _gpu(settings)
        plan = DevicePlan(gpu, batch)
        plan.run(torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev))
        parts = shard.dist_gather(plan.packed_summary(), 0)
        plan.check()
        assert len(parts) == 1
        got = unpack_summary(parts[0].cpu().numpy(), plan.max_k)
        _same(got, gpu.score_batch(batch))
    finally:
        dist.destroy_process_group()


def test_score_of_a_psm_too_big_for_one_workgroup_takes_the_batch_path():
    """C(26,4) = 14 950 site assignments (inside the fast kernels' limit of 15 000): the one-PSM kernel has no room
    for its sort area in one workgroup's LDS and declines (r03 advisor: it raised "LDS budget exceeded"); score()
    then goes through the plan machinery and answers like the reference."""
    rng = np.random.default_rng(12)
    pep = list("AKLGEDNVQR" * 4)
    for p in rng.permutation(40)[:26]:
        pep[p] = "STY"[p % 3]
    pep = "".join(pep)
    mz = np.sort(rng.uniform(150.0, 4200.0, 900))
    it = rng.lognormal(5, 1, 900)
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.05, fragment_types="by",
                    neutral_losses=[])
    g, c = _gpu(settings), _checker(settings)
    for s in (g, c):
        s.score(mz, it, pep, 4, 1)
    assert g.best_score == c.best_score and g.best_sequence == c.best_sequence
    assert np.array_equal(g.ascores, c.ascores)
    assert [a.tolist() for a in g.alt_sites] == [a.tolist() for a in c.alt_sites]
    assert len(g.pep_scores) == 14950


@pytest.mark.parametrize("mz_error", [0.05, 0.02, 0.004, 0.3])
def test_peaks_at_the_window_edges(mz_error):
    """Spectra whose peaks sit AT the window edges of the fragments -- the theoretical ions of random site assignments
    (float32 running sums as the scorer makes them) shifted by the tolerance plus or minus anything from one float32
    ulp to 0.05 -- must score exactly as the reference does: the window test is `f32(f - err) < peak < f32(f + err)` on
    each walker's own running sum, and a shortcut that decides a fragment for several site assignments at once (r04
    tried one: DESIGN.md section 10) has to get exactly these spectra right."""
    settings = dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=mz_error, fragment_types="by",
                    neutral_losses=[])
    batch, _ = synth.make_batch("cfg2", n_psm=400, seed=31)
    rng = np.random.default_rng(17)
    res = synth.RESIDUE_MASS
    mzs, its, offs = [], [], [0]
    for i in range(batch["n_psm"]):
        pep = bytes(batch["pep"][batch["pep_off"][i]:batch["pep_off"][i + 1]]).decode()
        sites = [p for p, ch in enumerate(pep) if ch in "STY"]
        ions = []
        for _ in range(6):                                   # fragments of random site assignments, float32 sums as the scorer makes them
            mod = set(rng.choice(sites, size=int(batch["n_of_mod"][i]), replace=False).tolist()) if sites else set()
            for order, a_off in ((range(len(pep) - 1), 0.0), (range(len(pep) - 1, 0, -1), 18.010565)):
                run = np.float32(0.0)
                for p in order:
                    run = np.float32(run + np.float32(np.float32(res[pep[p]]) + (np.float32(79.966331) if p in mod else np.float32(0.0))))
                    ions.append(float(np.float32(float(run) + a_off + 1.007825)))
        ions = np.asarray(ions)
        side = rng.choice([-1.0, 1.0], ions.size)
        ulp = np.spacing(ions.astype(np.float32)).astype(np.float64)
        off = np.where(rng.random(ions.size) < 0.5, ulp * rng.integers(-40, 41, ions.size), rng.uniform(-0.05, 0.05, ions.size))
        m = np.concatenate([ions + side * mz_error + off, batch["mz"][batch["peak_off"][i]:batch["peak_off"][i + 1]][::3]])
        m = np.sort(m[m > 50.0])
        mzs.append(m)
        its.append(rng.lognormal(5.0, 1.0, m.size))
        offs.append(offs[-1] + m.size)
    b2 = dict(batch, mz=np.concatenate(mzs), intensity=np.concatenate(its), peak_off=np.asarray(offs, np.int64))
    gpu, chk = _gpu(settings), _checker(settings)
    want = chk.score_batch(b2, int(b2["n_of_mod"].max()))
    got = gpu.score_batch(b2)
    for key in want:
        assert np.array_equal(got[key], want[key]), (mz_error, key)


def test_timing_events_of_runs_enqueued_back_to_back():
    """PYA_FLAG_TIMING keeps the events of the last 128 runs: pya_plan_timings_sum answers for every run since it was
    last asked (bench.py enqueues a block of steps and reads afterwards), a kernel family that launched nothing costs
    no event and reads 0, and the results of such a plan are the plain ones."""
    import torch
    from pyascore_amd.device import DevicePlan, unpack_summary
    batch, settings = synth.make_batch("cfg2", n_psm=3000, seed=77)
    gpu = _gpu(settings)
    dev = torch.device("cuda", 0)
    mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
    plan = DevicePlan(gpu, batch, timing=True)
    ms, n = plan.timings_sum()
    assert n == 0 and ms == (0.0, 0.0, 0.0, 0.0)
    for _ in range(7):
        plan.run(mz, it)
    ms, n = plan.timings_sum()
    assert n == 7
    assert ms[0] > 0 and ms[2] > 0                  # binning, the fused kernel
    assert ms[1] == 0 and ms[3] == 0                # nothing of cfg2 goes through the other two families
    last = plan.timings_ms()
    assert last[0] > 0 and last[2] > 0 and last[1] == 0 and last[3] == 0
    assert plan.timings_sum()[1] == 0
    for _ in range(150):                            # more runs than the ring holds: the latest 128 count
        plan.run(mz, it)
    ms, n = plan.timings_sum()
    assert n == 128 and ms[0] > 0
    got = unpack_summary(plan.packed_summary().cpu().numpy(), plan.max_k)
    plan.check()
    _same(got, gpu.score_batch(batch))
    plan.close()


def test_gather_records_are_packed_by_the_library():
    """pya_pack_records (the step's one pack kernel, straight into a slice of the send buffer) writes what the framework's
    concatenation of the result arrays wrote before -- for a plan at its own width, for a job-wide width wider than the batch
    needs (cfg3: 1..4 modifications packed at 6), into the middle of a larger buffer, and refuses a narrower one."""
    import torch
    from pyascore_amd.device import DevicePlan
    dev = torch.device("cuda", 0)
    for cfg, n, wide in (("cfg2", 1000, None), ("cfg3", 3000, 6), ("cfg5", 64, None)):
        batch, settings = synth.make_batch(cfg, n_psm=n, seed=88)
        gpu = _gpu(settings)
        mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
        plan = DevicePlan(gpu, batch, max_k=wide)
        plan.run(mz, it)
        k = plan.max_k
        cat = torch.cat([plan.best_score.view(torch.int32).unsqueeze(1), plan.n_sig.unsqueeze(1), plan.best_sig.view(torch.int32).view(-1, 2),
                         plan.ascores.view(torch.int32), plan.alt_mask.view(torch.int32).view(-1, 2 * k)], dim=1)
        assert torch.equal(plan.packed_summary(), cat)
        send = torch.full((n + 100, 4 + 3 * k), -7, dtype=torch.int32, device=dev)
        plan.packed_summary(out=send[50:50 + n])
        assert torch.equal(send[50:50 + n], cat) and bool((send[:50] == -7).all()) and bool((send[50 + n:] == -7).all())
        with pytest.raises(ValueError):
            plan.packed_summary(out=torch.empty((n, 3 + 3 * k), dtype=torch.int32, device=dev))
        plan.check()
        plan.close()


def test_hand_over_counts_between_the_runs_of_a_plan(monkeypatch):
    """A plan's runs alternate between two sets of hand-over counters, and the binning kernel of a run zeroes the set
    of the next (host_plan.cpp, bin_spectra.hip) -- no memset in between.  Ten runs of one plan whose every spectrum
    goes to the exact binning kernel (equal intensities everywhere) and whose every PSM is handed over by the fused
    kernel (PYA_DEBUG=512) must give the same, right results every time: counters that were not zeroed would grow
    past their lists."""
    import torch
    from pyascore_amd.device import DevicePlan, unpack_summary
    batch, settings = synth.make_batch("cfg2", n_psm=1500, seed=91)
    batch = dict(batch, intensity=np.round(batch["intensity"] / 50.0) * 50.0 + 50.0)      # count-like: ties in every window
    monkeypatch.setenv("PYA_DEBUG", "512")
    monkeypatch.setenv("PYA_NO_TINY", "1")
    gpu = _gpu(settings)
    want = _checker(settings).score_batch(batch, int(batch["n_of_mod"].max()))
    dev = torch.device("cuda", 0)
    mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
    plan = DevicePlan(gpu, batch)
    for run in range(10):
        plan.run(mz, it)
        if run in (0, 1, 2, 9):
            got = unpack_summary(plan.packed_summary().cpu().numpy(), plan.max_k)
            plan.check()
            for key in ("n_sig", "best_sig", "best_score", "ascores", "alt_mask"):
                assert np.array_equal(got[key], want[key]), (run, key)
    plan.close()
    monkeypatch.delenv("PYA_DEBUG", raising=False)
    switches.from_env(gpu)


def test_one_psm_stage_clocks():
    """pya_one_times: per-stage averages of the pya_score_one calls since it was last asked, on the host and inside
    the kernel (diagnostics behind scripts/one_probe.py)."""
    import ctypes as C
    batch, settings = synth.make_batch("cfg2", n_psm=12, seed=3)
    gpu = _gpu(settings)
    us = (C.c_double * 12)()
    gpu._lib.pya_one_times(gpu._h, C.byref(us))
    for i in range(12):
        gpu.score(**synth.unpack_psm(batch, i))
    assert gpu._lib.pya_one_times(gpu._h, C.byref(us)) == 0
    assert us[5] == 12
    assert all(us[i] > 0 for i in (2, 3, 6, 7, 8, 10)), list(us)
    assert 0.5e3 < us[10] / sum(us[6:10]) < 3.5e3       # cycles per microsecond: a shader clock between 0.5 and 3.5 GHz
    gpu._lib.pya_one_times(gpu._h, C.byref(us))
    assert us[5] == 0


def test_skip_invalid_sets_psms_aside():
    """PYA_FLAG_SKIP_INVALID: PSMs the library cannot take (a limit the reference does not have, an
    unknown residue, an empty spectrum, a spectrum without m/z windows) get a status code and
    "no result" rows; every other PSM of the batch scores exactly as it does alone."""
    good, settings = synth.make_batch("cfg3", n_psm=40, seed=8)
    rng = np.random.default_rng(4)
    mzs = np.sort(rng.uniform(150.0, 1500.0, 200))
    its = rng.lognormal(5, 1, 200)
    odd = [
        dict(mz=mzs, intensity=its, peptide="A" * 500 + "STY" * 20, n_of_mod=2, max_charge=1),         # 560 residues
        dict(mz=mzs, intensity=its, peptide="PEPTIXDESK", n_of_mod=1, max_charge=1),                     # unknown residue
        dict(mz=np.zeros(0), intensity=np.zeros(0), peptide="PEPTIDESK", n_of_mod=1, max_charge=1),      # empty spectrum
        dict(mz=mzs, intensity=its, peptide="ST" * 20, n_of_mod=12, max_charge=1),                       # C(40,12) assignments
        dict(mz=np.array([500.0]), intensity=np.array([1.0]), peptide="PEPTIDESK", n_of_mod=1, max_charge=1),   # no window
        dict(mz=mzs, intensity=its, peptide="PEPTIDESK", n_of_mod=1, max_charge=0),                      # charge 0
    ]
    psms = [dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"],
                 max_charge=kw["max_fragment_charge"]) for kw in (synth.unpack_psm(good, i) for i in range(40))]
    where = [3, 11, 12, 25, 33, 45]
    for w, o in zip(where, odd):
        psms.insert(w, o)
    batch = synth.pack_batch(psms)
    gpu = _gpu(settings)
    with pytest.raises(ValueError):
        gpu.score_batch(batch)
    got = gpu.score_batch(batch, skip_invalid=True)
    keep = np.setdiff1d(np.arange(batch["n_psm"]), where)
    assert list(got["status"][where]) == [17, 16, 16, 17, 1, 16] and not got["status"][keep].any()
    assert "PSM 3" in got["status_message"]
    assert np.all(got["best_score"][where] == -1.0) and np.all(got["n_sig"][where] == -1)
    assert not got["ascores"][where].any() and not got["alt_mask"][where].any()
    alone = gpu.score_batch(good)
    k = alone["ascores"].shape[1]
    for key in ("n_sig", "best_sig", "best_score"):
        assert np.array_equal(got[key][keep], alone[key]), key
    assert np.array_equal(got["ascores"][keep][:, :k], alone["ascores"]) and not got["ascores"][keep][:, k:].any()
    assert np.array_equal(got["alt_mask"][keep][:, :k], alone["alt_mask"])
    # a clean batch reports all zeros
    assert not gpu.score_batch(good, skip_invalid=True)["status"].any()


def _long_batch(L, n_sites, n_mod, n, seed, **over):
    return synth.make_batch("cfg2", n_psm=n, seed=seed, L=L, n_sites=n_sites, n_mod=n_mod, **over)


def _same_psm_by_psm(gpu, chk, batch, records=True):
    """score() + every property, PSM by PSM (the batch form of the checker packs alternative sites into 64-bit residue
    masks, which peptides of more than 64 residues do not fit), and the retained records in one bulk call per side."""
    for i in range(batch["n_psm"]):
        kw = synth.unpack_psm(batch, i)
        gpu.score(**kw)
        chk.score(**kw)
        assert gpu.best_sequence == chk.best_sequence, i
        assert np.float32(gpu.best_score) == np.float32(chk.best_score), i
        assert np.array_equal(gpu.ascores, chk.ascores), i
        ga, ca = gpu.alt_sites, chk.alt_sites
        assert len(ga) == len(ca) and all(np.array_equal(a, b) for a, b in zip(ga, ca)), i
        if records:
            raw = chk.raw_pep_scores()
            bits = (raw["signature"].astype(np.uint64) << np.arange(raw["signature"].shape[1], dtype=np.uint64)).sum(axis=1)
            got = gpu.batch_pep_scores() if gpu._batch_n else {k: np.zeros(0) for k in ("sig_bits", "counts", "scores", "weighted_score", "total_fragments")}
            if gpu._batch_n is None:
                gpu._ensure_kept()
                got = gpu.batch_pep_scores()
            assert np.array_equal(got["sig_bits"], bits.astype(np.uint64)), i
            for key in ("counts", "scores", "weighted_score", "total_fragments"):
                assert np.array_equal(got[key], raw[key]), (i, key)


@pytest.mark.parametrize("L,n_sites,n_mod,over", [
    (65, 4, 2, {}), (100, 6, 2, dict(max_charge=2)), (180, 5, 3, {}), (255, 4, 1, {}),
    (256, 5, 2, {}), (400, 4, 2, dict(max_charge=2)), (511, 6, 3, {}),
    (90, 5, 2, dict(fragment_types="yb", mz_error=0.5)),
    (70, 4, 2, dict(fragment_types="bycz", max_charge=2, mz_error=0.02, neutral_loss=("sty", 97.9769))),
])
def test_general_kernel_long_peptides(L, n_sites, n_mod, over):
    """Peptides of 65 to 511 residues (the reference takes any length: cpp/ModifiedPeptide.cpp:24-57) are scored by the
    general kernel (csrc/general_psm.hip), whole: counts, PepScores, the sorted order, Ascores and alternative sites
    equal the reference's."""
    batch, settings = _long_batch(L, n_sites, n_mod, 6, 1000 + L, **over)
    _same_psm_by_psm(_gpu(settings), _checker(settings), batch)


def test_general_kernel_many_site_assignments_and_long_lists():
    """More than 15 000 site assignments (C(20,5) = 15 504, C(17,8) = 24 310: the std::sort emulation in global memory,
    32-bit positions) and more than 2 048 fragments per ion type (neutral losses x eight charges) -- in one batch with
    ordinary PSMs, which keep their fast kernels; then the big ones' retained records."""
    big1, settings = _long_batch(40, 20, 5, 3, 71)
    big2, _ = _long_batch(30, 17, 8, 2, 72)
    small, _ = synth.make_batch("cfg2", n_psm=40, seed=73)
    psms = []
    for bt in (small, big1, big2):
        for i in range(bt["n_psm"]):
            kw = synth.unpack_psm(bt, i)
            psms.append(dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"],
                             max_charge=kw["max_fragment_charge"]))
    rng = np.random.default_rng(3)
    order = rng.permutation(len(psms))
    batch = synth.pack_batch([psms[i] for i in order])
    gpu, chk = _gpu(settings), _checker(settings)
    got = gpu.score_batch(batch, skip_invalid=True)
    assert not got["status"].any()
    want = chk.score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        assert np.array_equal(got[key], want[key]), key
    assert sorted(np.unique(got["n_sig"])) == [20, 15504, 24310]
    _same_psm_by_psm(gpu, chk, big1)
    # long fragment lists: two neutral losses and eight fragment charges on one ion type
    st = dict(settings, fragment_types="b", mz_error=0.02, neutral_losses=[["sty", 97.9769], ["ST", 18.01528]])
    lists, _ = synth.make_batch("cfg2", n_psm=4, seed=74, L=60, n_sites=5, n_mod=2, max_charge=8)
    _same_psm_by_psm(_gpu(st), _checker(st), lists)


@pytest.mark.parametrize("n_masses", [5, 6, 8])
def test_more_than_four_neutral_loss_masses(n_masses):
    """The reference takes any number of neutral-loss entries (cpp/ModifiedPeptide.cpp:99-103); the fast kernels pack the
    loss state of a fragment into 8 bits = four distinct masses.  With five to eight distinct masses every PSM of the scorer
    goes through the general kernel (csrc/general_psm.hip: a 16-bit state, up to 45 distinct sums of at most two losses):
    same results, records and calculate_ambiguity included.  A ninth distinct mass is refused."""
    from pyascore_amd import PyAscore
    losses = [["st", 97.9769], ["y", 79.9663], ["ST", 18.01528], ["D", 18.0106], ["E", 17.0265], ["K", 17.0265 + 1.0], ["R", 43.99],
              ["N", 17.5]][:n_masses]
    # (the score table covers 4 096 trials = (L - 1) x charges x loss sums x ion types at most: 19 x 2 x 45 x 2, 19 x 1 x 28 x 4)
    batch, settings = synth.make_batch("cfg2", n_psm=5, seed=600 + n_masses, L=20, n_sites=5, n_mod=2, max_charge=1 if n_masses == 6 else 2)
    st = dict(settings, fragment_types="bycz" if n_masses == 6 else "by", mz_error=0.02, neutral_losses=losses)
    gpu, chk = _gpu(st), _checker(st)
    _same_psm_by_psm(gpu, chk, batch)
    got = gpu.score_batch(batch)
    want = chk.score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        assert np.array_equal(got[key], want[key]), key
    for i in range(2):
        _ambiguity_agrees(gpu, chk, synth.unpack_psm(batch, i))
    if n_masses == 8:
        with pytest.raises(ValueError, match="neutral-loss"):
            gpu.add_neutral_loss("Q", 1.2345)
        gpu.score(**synth.unpack_psm(batch, 0))             # (the refused call left the settings as they were)
        chk.score(**synth.unpack_psm(batch, 0))
        assert gpu.best_sequence == chk.best_sequence and np.array_equal(gpu.ascores, chk.ascores)


def _ambiguity_agrees(gpu, chk, kw, picks=(1, 2, -1)):
    """calculate_ambiguity of the last scored PSM for a few record pairs, both ways round, against the checker."""
    gpu.score(**kw)
    chk.score(**kw)
    ps, cps = gpu.pep_scores, chk.pep_scores
    assert len(ps) == len(cps) and len(ps) > 1
    assert gpu.calculate_ambiguity(ps[0], ps[0]) == 0.0
    seen = 0
    for j in picks:
        j = j % len(ps)
        if j == 0:
            continue
        assert np.array_equal(ps[j]["signature"], cps[j]["signature"])
        assert gpu.calculate_ambiguity(ps[0], ps[j]) == chk.calculate_ambiguity(cps[0], cps[j]), j
        assert gpu.calculate_ambiguity(ps[j], ps[0]) == chk.calculate_ambiguity(cps[j], cps[0]), j
        seen += 1
    assert seen
    # the value for the runner-up of a site is that site's Ascore (Ascore.cpp:239-251)
    assert np.any(np.isclose(gpu.calculate_ambiguity(ps[0], ps[1]), gpu.ascores)) or len(ps) > 2


def test_calculate_ambiguity_beyond_the_fast_kernel():
    """PyAscore.calculate_ambiguity (Ascore.pyx:208-230, cpp/Ascore.cpp:157-210) wherever the reference computes it: after
    score() of a spectrum of more than 8 192 peaks (its retained table is larger than the LDS the fast kernel is launched
    with: r04 advisor finding), of a peptide of more than 64 residues, with n_top 11 / 16 (n_top depth scores per
    container, all of them searched for the depth), and with neutral losses + several charges on a long peptide."""
    rng = np.random.default_rng(5)
    small, settings = synth.make_batch("cfg2", n_psm=4, seed=91)
    gpu, chk = _gpu(settings), _checker(settings)
    for j, P in enumerate((8193, 20000, 65535)):                       # big spectra
        kw = synth.unpack_psm(small, j)
        mz = np.concatenate([kw["mz_arr"], rng.uniform(100.0, 2500.0, P - kw["mz_arr"].size)])
        it = np.concatenate([kw["int_arr"], rng.lognormal(4.0, 1.0, P - kw["int_arr"].size)])
        o = np.argsort(mz, kind="stable")
        _ambiguity_agrees(gpu, chk, dict(kw, mz_arr=mz[o], int_arr=it[o]))
    _ambiguity_agrees(gpu, chk, synth.unpack_psm(small, 3))          # (an ordinary PSM afterwards: the fast kernel again)
    for L, n_sites, n_mod, over in ((65, 4, 2, {}), (130, 6, 3, dict(max_charge=2)), (255, 5, 2, {}), (511, 4, 2, {}),
                                    (70, 4, 2, dict(fragment_types="bycz", max_charge=2, mz_error=0.02,
                                                    neutral_loss=("sty", 97.9769)))):
        batch, st = _long_batch(L, n_sites, n_mod, 2, 2000 + L, **over)
        g2, c2 = _gpu(st), _checker(st)
        for i in range(batch["n_psm"]):
            _ambiguity_agrees(g2, c2, synth.unpack_psm(batch, i))
    for n_top in (11, 16):
        for cfg in ("cfg2", "cfg4"):
            batch, st = synth.make_batch(cfg, n_psm=3, seed=950 + n_top)
            st = dict(st, n_top=n_top)
            g2, c2 = _gpu(st), _checker(st)
            for i in range(batch["n_psm"]):
                _ambiguity_agrees(g2, c2, synth.unpack_psm(batch, i))
            ps = g2.pep_scores
            assert ps[0]["scores"].shape == (n_top,)
            with pytest.raises(ValueError, match="depth scores"):
                g2.calculate_ambiguity(dict(ps[0], scores=ps[0]["scores"][:10]), ps[1])


def test_one_huge_psm_does_not_size_everybody_elses_scratch():
    """The general kernel's scratch (sort area of the std::sort emulation, competitor list) is a slice per PSM, sized for
    that PSM: a batch of a thousand small PSMs and ONE with C(20,10) = 184 756 site assignments on a scorer with n_top = 12
    (every PSM through the general kernel) needs megabytes, not a thousand times the big one's room (r04 advisor
    finding: it was sized by the largest), and pya_score_batch's chunking accounts for it."""
    from pyascore_amd import PyAscore
    from pyascore_amd.device import DevicePlan
    small, settings = synth.make_batch("cfg2", n_psm=1000, seed=61)
    big, _ = _long_batch(30, 20, 10, 1, 62)
    st = dict(settings, n_top=12)
    psms = []
    for bt in (small, big):
        for i in range(bt["n_psm"]):
            kw = synth.unpack_psm(bt, i)
            psms.append(dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"],
                             max_charge=kw["max_fragment_charge"]))
    batch = synth.pack_batch(psms)
    gpu, chk = _gpu(st), _checker(st)
    plan = DevicePlan(gpu, batch)
    assert plan.workspace_bytes < 64 << 20, plan.workspace_bytes          # (sized by the largest: 1001 x 1.5 MB)
    plan.close()
    got = gpu.score_batch(batch)
    want = chk.score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        assert np.array_equal(got[key], want[key]), key
    assert int(got["n_sig"][-1]) == 184756


def test_fragment_charges_above_sixteen():
    """max_fragment_charge 17 ... 40 (the reference takes any charge, cpp/ModifiedPeptide.cpp:81-97; r04 refused above 16):
    short peptides on the fast kernels while their fragment lists fit, the general kernel beyond."""
    for L, n_sites, n_mod, z, over in ((9, 4, 2, 17, {}), (12, 5, 2, 24, {}), (8, 3, 1, 40, dict(fragment_types="bycz", mz_error=0.02)),
                                       (30, 6, 3, 20, {})):
        batch, settings = synth.make_batch("cfg2", n_psm=8, seed=300 + z, L=L, n_sites=n_sites, n_mod=n_mod, max_charge=z, **over)
        gpu, chk = _gpu(settings), _checker(settings)
        got = gpu.score_batch(batch)
        want = chk.score_batch(batch, got["ascores"].shape[1])
        for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
            assert np.array_equal(got[key], want[key]), (z, key)
        _same_psm_by_psm(gpu, chk, synth.slice_batch(batch, 0, 2))
    kw = synth.unpack_psm(batch, 0)
    with pytest.raises(ValueError, match="max_fragment_charge"):
        gpu.score(**dict(kw, max_fragment_charge=256))


def test_spectra_of_more_than_8192_peaks():
    """8 193 to 65 535 peaks: binned by pya_bin_global_kernel (the general binning body with its arrays in the workspace),
    scored by the general kernel; sorted, unsorted and tie-heavy, next to ordinary PSMs in one batch."""
    rng = np.random.default_rng(23)
    small, settings = synth.make_batch("cfg2", n_psm=12, seed=81)
    psms = []
    for i in range(small["n_psm"]):
        kw = synth.unpack_psm(small, i)
        psms.append(dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"], n_of_mod=kw["n_of_mod"], max_charge=1))
    for j, (P, mode) in enumerate([(8193, "sorted"), (20000, "sorted"), (65535, "sorted"), (9000, "shuffled"), (12000, "counts")]):
        base = psms[j]
        mz = np.concatenate([base["mz"], rng.uniform(100.0, 2500.0, P - base["mz"].size)])
        it = np.concatenate([base["intensity"], rng.lognormal(4.0, 1.0, P - base["intensity"].size)])
        o = np.argsort(mz, kind="stable")
        mz, it = mz[o], it[o]
        if mode == "shuffled":
            q = rng.permutation(P)
            mz, it = mz[q], it[q]
        if mode == "counts":
            it = np.floor(it / np.median(it) * 5.0) + 1.0
        psms.append(dict(base, mz=mz, intensity=it))
    batch = synth.pack_batch(psms)
    gpu, chk = _gpu(settings), _checker(settings)
    got = gpu.score_batch(batch, skip_invalid=True)
    assert not got["status"].any()
    want = chk.score_batch(batch, got["ascores"].shape[1])
    for key in want:
        assert np.array_equal(got[key], want[key]), key
    kw = synth.unpack_psm(batch, batch["n_psm"] - 4)           # 20 000 peaks through score()
    gpu.score(**kw)
    chk.score(**kw)
    assert gpu.best_sequence == chk.best_sequence and np.float32(gpu.best_score) == np.float32(chk.best_score)
    assert np.array_equal(gpu.ascores, chk.ascores)


@pytest.mark.parametrize("n_top", [11, 12, 16])
def test_n_top_above_ten(n_top):
    """n_top > 10: the reference retains n_top peaks per window, counts and scores n_top depths, weights the first ten and
    searches all of them for the depth of an Ascore (cpp/Spectra.cpp:24-41, cpp/Ascore.cpp:15-36, :123-139, :164-172).
    Such a scorer sends every PSM through the general kernel; results, records (n_top wide) and the tie-heavy binning
    equal the reference's.  Below 10 the reference reads past its scores: refused."""
    from pyascore_amd import PyAscore
    for cfg, n, over in (("cfg2", 24, {}), ("cfg3", 30, {}), ("cfg4", 4, {})):
        batch, settings = synth.make_batch(cfg, n_psm=n, seed=900 + n_top, **over)
        st = dict(settings, n_top=n_top)
        gpu, chk = _gpu(st), _checker(st)
        got = gpu.score_batch(batch)
        want = chk.score_batch(batch, got["ascores"].shape[1])
        for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
            assert np.array_equal(got[key], want[key]), (cfg, key)
        _same_psm_by_psm(gpu, chk, synth.slice_batch(batch, 0, 3))
    # count-like intensities: ties at the top of the windows decide which n_top peaks stay
    batch, settings = synth.make_batch("cfg2", n_psm=40, seed=77)
    tied = dict(batch, intensity=np.floor(batch["intensity"] / np.median(batch["intensity"]) * 6.0) + 1.0)
    st = dict(settings, n_top=n_top)
    got = _gpu(st).score_batch(tied)
    want = _checker(st).score_batch(tied, got["ascores"].shape[1])
    for key in want:
        assert np.array_equal(got[key], want[key]), ("tied", key)
    with pytest.raises(ValueError, match="n_top"):
        PyAscore(100.0, 9, "STY", 79.966331)
    with pytest.raises(ValueError, match="n_top"):
        PyAscore(100.0, 17, "STY", 79.966331)


def test_retained_batch_invalidates_single_psm_state():
    """score() then score_batch(keep=True): the handle's retained records now belong to the batch, so
    the per-PSM properties go back to their initial state instead of mixing the two."""
    batch, settings = synth.make_batch("cfg2", n_psm=8, seed=12)
    gpu = _gpu(settings)
    gpu.score(**synth.unpack_psm(batch, 5))
    assert gpu.best_sequence and len(gpu.pep_scores) == 20
    gpu.score_batch(batch, keep=True)
    assert gpu.best_sequence == "" and gpu.best_score == -1.0 and gpu.pep_scores == [] and gpu.ascores.size == 0
    with pytest.raises(RuntimeError):
        gpu.calculate_ambiguity({}, {})
    assert gpu.batch_pep_scores()["rec_off"][-1] == 8 * 20
    gpu.score(**synth.unpack_psm(batch, 5))
    assert len(gpu.pep_scores) == 20


def test_lazy_records_belong_to_the_psm_that_was_scored():
    """score(A) produces A's per-signature records only when pep_scores is read (a replay of what score() staged in
    the library).  Whatever reuses that staging in between -- a batch of one PSM B, a change of the settings, a reload
    of the switches -- must not make the replay return another PSM's records, or A's under other settings."""
    batch, settings = synth.make_batch("cfg3", n_psm=24, seed=31)
    chk = _checker(settings)

    def want_scores(i):
        chk.score(**synth.unpack_psm(batch, i))
        return [(int(np.dot(r["signature"], 1 << np.arange(len(r["signature"])))), float(r["weighted_score"]),
                 list(map(int, r["counts"]))) for r in chk.pep_scores]

    def got_scores(g):
        return [(int(np.dot(r["signature"], 1 << np.arange(len(r["signature"])))), float(r["weighted_score"]),
                 list(map(int, r["counts"]))) for r in g.pep_scores]

    a, bi = 3, 17
    one_b = {k: v for k, v in synth.make_slice_of(batch, bi, bi + 1).items()} if hasattr(synth, "make_slice_of") else None
    if one_b is None:                                       # a batch holding PSM `bi` alone
        po, pe = batch["peak_off"], batch["pep_off"]
        one_b = dict(n_psm=1, mz=batch["mz"][po[bi]:po[bi + 1]].copy(), intensity=batch["intensity"][po[bi]:po[bi + 1]].copy(),
                     peak_off=np.array([0, po[bi + 1] - po[bi]], np.int64), pep=batch["pep"][pe[bi]:pe[bi + 1]].copy(),
                     pep_off=np.array([0, pe[bi + 1] - pe[bi]], np.int64), n_of_mod=batch["n_of_mod"][bi:bi + 1].copy(),
                     max_charge=batch["max_charge"][bi:bi + 1].copy(), aux_pos=np.zeros(0, np.uint32),
                     aux_mass=np.zeros(0, np.float32), aux_off=np.zeros(2, np.int64))
    gpu = _gpu(settings)
    gpu.score(**synth.unpack_psm(batch, a))
    out = gpu.score_batch(one_b)                            # reuses the one-PSM path of the library
    assert out["n_sig"][0] == len(want_scores(bi))
    assert got_scores(gpu) == want_scores(a)                # A's records, not B's read with A's shape
    # ... and after a change of the settings the records are still those of the settings A was scored with
    gpu.score(**synth.unpack_psm(batch, a))
    gpu.add_neutral_loss("STY", 97.9769)
    assert got_scores(gpu) == want_scores(a)
    # the C ABI itself: a replay after the staging was reused fails instead of answering for the wrong PSM
    from pyascore_amd import _lib
    gpu2 = _gpu(settings)
    gpu2.score(**synth.unpack_psm(batch, a))
    gpu2._last["lazy"] = False                              # (skip the Python side's own precaution)
    gpu2.score_batch(one_b)
    assert gpu2._lib.pya_rescore_last_keep(gpu2._h) == _lib.PYA_ERR_STATE


def test_drop_in_import_name():
    """`from pyascore import PyAscore` (reference pyascore/__init__.py:17) gives the HIP scorer."""
    import pyascore
    import pyascore_amd
    assert pyascore.PyAscore is pyascore_amd.PyAscore
    s = pyascore.PyAscore(bin_size=100., n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.05)
    mz = np.array([150.0, 300.5, 420.25]); it = np.array([1.0, 2.0, 3.0])
    s.score(mz, it, "PEPTIDE", 1)
    assert s.best_sequence == "PEPT[80]IDE"


@pytest.mark.parametrize("case", ["velos_zprec", "edge_nKc", "edge_default", "synth_cfg3"])
def test_bulk_sequence_strings(case):
    """pya_format_peptides: every best_sequence of a batch, and the sequence of every pep_scores record,
    in one library call each -- equal to the reference's strings (ModifiedPeptide.cpp:199-253)."""
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    gpu = _gpu(settings)
    res = gpu.score_batch(batch, keep=True)
    assert gpu.format_batch(batch, res["best_sig"], valid=res["n_sig"]) == expected["best_sequence"].tolist()
    bulk = gpu.batch_pep_scores(batch=batch)
    assert len(bulk["sequence"]) == int(bulk["rec_off"][-1])
    for i in range(0, batch["n_psm"], max(1, batch["n_psm"] // 6)):
        gpu.score(**synth.unpack_psm(batch, i))
        want = [p["sequence"] for p in gpu.pep_scores]
        gpu.score_batch(batch, keep=True)
        assert bulk["sequence"][int(bulk["rec_off"][i]):int(bulk["rec_off"][i + 1])] == want
    with pytest.raises(ValueError):
        gpu.format_batch(batch, res["best_sig"], rec_psm=np.full(batch["n_psm"], batch["n_psm"], np.int64))


def test_retained_export_of_a_batch_over_the_budget():
    """score_batch(keep=True) of a batch whose records do not fit the workspace budget: scored without
    them, and batch_pep_scores() re-scores the range it is asked for a budget's worth at a time -- the
    same records as the one-plan export, for the whole batch and for a range inside it."""
    batch, settings = synth.make_batch("cfg3", n_psm=3000, seed=31)
    gpu = _gpu(settings)
    res = gpu.score_batch(batch, keep=True)
    assert gpu._lazy_batch is None                     # fits the default budget: one retained plan
    want = gpu.batch_pep_scores(batch=batch)
    gpu.set_workspace_budget(16 << 20)                 # the records of 3000 PSMs need more than that
    res2 = gpu.score_batch(batch, keep=True)
    assert gpu._lazy_batch is not None
    for key in res:
        assert np.array_equal(res[key], res2[key]), key
    got = gpu.batch_pep_scores(batch=batch)
    for key in ("rec_off", "sig_bits", "counts", "scores", "weighted_score", "total_fragments"):
        assert np.array_equal(got[key], want[key]), key
    assert got["sequence"] == want["sequence"]
    part = gpu.batch_pep_scores(1200, 1900)
    a, b = int(want["rec_off"][1200]), int(want["rec_off"][1900])
    assert np.array_equal(part["rec_off"], want["rec_off"][1200:1901] - a)
    assert np.array_equal(part["sig_bits"], want["sig_bits"][a:b]) and np.array_equal(part["scores"], want["scores"][a:b])
    gpu.set_workspace_budget(0)


def test_chunked_calls_equal_one_plan(monkeypatch):
    """pya_score_batch cuts big calls into chunks that fit the device budget and pipelines them
    (upload of chunk c + 1 under the kernels of chunk c): same results as one plan for the whole
    batch, whatever the cut; errors and set-aside PSMs keep their index in the caller's batch."""
    batch, settings = synth.make_slice(synth.describe("cfg3", 60_000, seed=17)), None
    settings = synth.describe("cfg3", 1, seed=17)["settings"]
    gpu = _gpu(settings)
    monkeypatch.setenv("PYA_NO_CHUNKS", "1")
    switches.from_env(gpu)                                   # (the switches are read once per scorer)
    one = gpu.score_batch(batch)
    monkeypatch.delenv("PYA_NO_CHUNKS")
    switches.from_env(gpu)
    _same(gpu.score_batch(batch), one)                 # default: ~96 MB of spectra per chunk
    gpu.set_workspace_budget(48 << 20)                 # tight budget: dozens of chunks
    _same(gpu.score_batch(batch), one)
    monkeypatch.setenv("PYA_CHUNK_MB", "3")            # ... and ~100 chunks of 3 MB
    switches.from_env(gpu)
    _same(gpu.score_batch(batch), one)
    monkeypatch.delenv("PYA_CHUNK_MB")
    switches.from_env(gpu)
    gpu.set_workspace_budget(0)
    with pytest.raises(ValueError):
        gpu.set_workspace_budget(1000)
    # an invalid PSM deep inside the batch: reported with its index in the caller's batch
    bad = dict(batch, pep=batch["pep"].copy())
    where = 51_234
    bad["pep"][batch["pep_off"][where] + 1] = ord("X")
    with pytest.raises(ValueError, match="PSM %d: unknown residue" % where):
        gpu.score_batch(bad)
    got = gpu.score_batch(bad, skip_invalid=True)
    assert np.flatnonzero(got["status"]).tolist() == [where] and got["best_score"][where] == -1.0
    keep = np.arange(batch["n_psm"]) != where
    for key in ("n_sig", "best_sig", "best_score", "ascores", "alt_mask"):
        assert np.array_equal(got[key][keep], one[key][keep]), key


def test_one_call_of_a_million_heavy_psms():
    """1M PSMs of the cfg5 shape (3003 site assignments each: 86 GB of workspace as one plan) in ONE
    score_batch call under the default 6 GiB budget; a random sample agrees with the reference."""
    desc = synth.describe("cfg5", 1_000_000, seed=99)
    batch = synth.make_slice(desc)
    gpu = _gpu(desc["settings"])
    full = gpu.score_batch(batch)
    assert np.all(full["n_sig"] == 3003)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(batch["n_psm"], 100, replace=False))
    sub = synth.pack_batch([dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"],
                                 n_of_mod=kw["n_of_mod"], max_charge=kw["max_fragment_charge"])
                            for kw in (synth.unpack_psm(batch, int(i)) for i in pick)])
    want = _checker(desc["settings"]).score_batch(sub, 5)
    for key in ("n_sig", "best_sig", "best_score", "ascores", "alt_mask"):
        assert np.array_equal(full[key][pick], want[key]), key


def test_scorer_first_then_torch_in_one_process():
    """Import order must not matter: score through the library, THEN bring up torch's HIP runtime and run
    a device-resident plan on torch tensors in the same process."""
    import subprocess
    import sys
    code = """
import numpy as np
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan, unpack_summary
batch, st = synth.make_batch("cfg3", n_psm=400, seed=5)
s = PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], mz_error=st["mz_error"],
             fragment_types=st["fragment_types"])
want = s.score_batch(batch)
import torch
dev = torch.device("cuda", s.device)
plan = DevicePlan(s, batch)
plan.run(torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev))
plan.check()
got = unpack_summary(plan.packed_summary().cpu().numpy(), int(batch["n_of_mod"].max()))
assert np.array_equal(got["best_score"].view(np.uint32), want["best_score"].view(np.uint32))
assert np.array_equal(got["best_sig"], want["best_sig"])
print("same")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("same"), out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mod_mass,mz_error", [(57.02146, 0.05), (57.02146, 0.45), (114.04293, 0.3), (0.984016, 0.02),
                                                (79.966331, 0.49), (79.966331, 4.0), (57.02146, 4.0)])
def test_modification_mass_that_mimics_residues(mod_mass, mz_error, path):
    """Site-determining ions when moving the modification reproduces other fragments' m/z: a modification as
    heavy as glycine (or two of them, next to GG), and one lighter than a dalton (deamidation) with a
    tolerance that makes neighbouring lists overlap.  The fused kernel pairs ions only between the first and
    the last differing residue and relies on "at most one partner per ion" there; these are the inputs that
    stress both assumptions, on every route of the scorer."""
    rng = np.random.default_rng(int(mod_mass * 1000) + int(mz_error * 100))
    settings = dict(bin_size=100.0, n_top=10, mod_group="STYN", mod_mass=mod_mass, mz_error=mz_error,
                    fragment_types="by", neutral_losses=[])
    peps = ["SGSGTGYGK", "GSGGSGGTK", "NGSGNGTGYR", "AGSTGGYGSGK", "SGGGSAGTGNK", "GGSGGSGGTGGYK", "TGSGNR", "SGTK"]
    psms = []
    for rep in range(40):
        pep = peps[rep % len(peps)]
        n_sites = sum(ch in "STYN" for ch in pep)
        k = 1 + rep % min(3, n_sites - 1)
        # peaks: every b/y fragment of a random assignment, jittered inside and just outside the tolerance,
        # plus noise
        masses = dict(G=57.02146, S=87.03203, T=101.04768, Y=163.06333, N=114.04293, A=71.03711, K=128.09496, R=156.10111)
        sites = [i for i, ch in enumerate(pep) if ch in "STYN"]
        chosen = set(rng.choice(sites, size=k, replace=False).tolist())
        res = [masses[ch] + (mod_mass if i in chosen else 0.0) for i, ch in enumerate(pep)]
        frag = []
        run = 0.0
        for m in res[:-1]:
            run += m
            frag.append(run + 1.007825)
        run = 18.010565
        for m in res[::-1][:-1]:
            run += m
            frag.append(run + 1.007825)
        mz = np.array(frag)
        mz = np.concatenate([mz + rng.uniform(-1.2, 1.2, mz.size) * mz_error, rng.uniform(60.0, 1400.0, 30)])
        mz = np.sort(mz)
        psm = dict(mz=mz, intensity=rng.lognormal(5, 1, mz.size), peptide=pep, n_of_mod=k, max_charge=1)
        if rep % 5 == 4:
            # a fixed modification that leaves a glycine lighter than a dalton: neighbouring ions of one list
            # closer than two tolerances, the case the span shortcut must not be taken for
            psm["aux_pos"] = np.array([pep.index("G") + 1], np.uint32)
            psm["aux_mass"] = np.array([-56.42], np.float32)
        psms.append(psm)
    batch = synth.pack_batch(psms)
    got = _gpu(settings).score_batch(batch)
    want = _checker(settings).score_batch(batch, got["ascores"].shape[1])
    _same(got, want)

