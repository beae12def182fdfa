"""Host-side logic that needs no GPU: synthetic generator, CSR packing, shard partitioning,
summary record packing."""
import numpy as np
import pytest

from pyascore_amd import shard, synth


def test_generator_shapes_and_determinism():
    a, sa = synth.make_batch("cfg2", n_psm=50, seed=3)
    b, _ = synth.make_batch("cfg2", n_psm=50, seed=3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["n_psm"] == 50 and a["peak_off"].shape == (51,) and a["pep_off"][-1] == 50 * 20
    assert np.all(np.diff(a["pep_off"]) == 20) and np.all(a["n_of_mod"] == 3)
    peaks = np.diff(a["peak_off"])
    assert 300 <= peaks.min() and peaks.max() <= 338            # 300 noise + kept signal peaks
    for i in range(50):
        m = a["mz"][a["peak_off"][i]:a["peak_off"][i + 1]]
        assert np.all(np.diff(m) >= 0)
        pep = bytes(a["pep"][a["pep_off"][i]:a["pep_off"][i + 1]]).decode()
        assert sum(c in "STY" for c in pep) == 6
    assert sa["mz_error"] == 0.05 and sa["fragment_types"] == "by"


def test_cfg3_shapes():
    b, _ = synth.make_batch("cfg3", n_psm=400, seed=5)
    L = np.diff(b["pep_off"])
    assert L.min() >= 8 and L.max() <= 40
    assert b["n_of_mod"].min() >= 1 and b["n_of_mod"].max() <= 4
    for i in range(400):
        pep = bytes(b["pep"][b["pep_off"][i]:b["pep_off"][i + 1]]).decode()
        n_sites = sum(c in "STY" for c in pep)
        assert b["n_of_mod"][i] + 1 <= n_sites <= min(12, len(pep) - 1)


def test_pack_unpack_slice_roundtrip():
    b, _ = synth.make_batch("cfg3", n_psm=30, seed=8)
    psms = []
    for i in range(30):
        kw = synth.unpack_psm(b, i)
        psms.append(dict(mz=kw["mz_arr"], intensity=kw["int_arr"], peptide=kw["peptide"],
                         n_of_mod=kw["n_of_mod"], max_charge=kw["max_fragment_charge"]))
    p = synth.pack_batch(psms)
    for k in ("mz", "intensity", "peak_off", "pep", "pep_off", "n_of_mod", "max_charge"):
        assert np.array_equal(p[k], b[k]), k
    s = synth.slice_batch(b, 7, 19)
    assert s["n_psm"] == 12 and s["peak_off"][0] == 0 and s["pep_off"][0] == 0
    assert synth.unpack_psm(s, 0)["peptide"] == synth.unpack_psm(b, 7)["peptide"]
    assert np.array_equal(synth.unpack_psm(s, 11)["mz_arr"], synth.unpack_psm(b, 18)["mz_arr"])


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_is_contiguous_cover_and_balanced(world):
    b, _ = synth.make_batch("cfg3", n_psm=2000, seed=9)
    w = shard.work_estimate(b)
    ranges = shard.partition(w, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == 2000
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 <= a1
    loads = np.array([w[lo:hi].sum() for lo, hi in ranges])
    assert loads.max() <= w.sum() / world + w.max() + 1e-9


def test_work_estimate_matches_definition():
    b, _ = synth.make_batch("cfg5", n_psm=3, seed=1)
    w = shard.work_estimate(b)
    assert np.allclose(w, 3003 * 29 * 2 * 1)


def test_summary_pack_roundtrip():
    from pyascore_amd.device import unpack_summary
    rng = np.random.default_rng(0)
    n, k = 17, 3
    best_score = rng.random(n).astype(np.float32)
    n_sig = rng.integers(0, 100, n).astype(np.int32)
    best_sig = rng.integers(0, 2 ** 62, n).astype(np.uint64)
    asc = rng.standard_normal((n, k)).astype(np.float32)
    alt = rng.integers(0, 2 ** 62, (n, k)).astype(np.uint64)
    packed = np.concatenate([best_score.view(np.int32)[:, None], n_sig[:, None],
                             best_sig.view(np.int32).reshape(n, 2), asc.view(np.int32),
                             alt.view(np.int32).reshape(n, 2 * k)], axis=1)
    u = unpack_summary(packed, k)
    assert np.array_equal(u["best_score"], best_score) and np.array_equal(u["n_sig"], n_sig)
    assert np.array_equal(u["best_sig"], best_sig) and np.array_equal(u["ascores"], asc)
    assert np.array_equal(u["alt_mask"], alt)
