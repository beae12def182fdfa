"""score_big's count-node table (csrc/walk_core.hip.h): one peak lookup per (direction, step, number of modified residues
so far) decides the fragment for every site assignment through that node, the exact envelope of their float32 running
sums says when it cannot (a peak within a few ulps of a window end: the node is marked and the walkers look up
themselves), and a site assignment's counts are k table reads.  Everything the reference computes per site assignment
(cpp/Ascore.cpp:53-139: counts, PepScores, their order) must come out bit-equal -- with the table, without it
(PYA_DEBUG=0x8000: the two-level prefix tree and a lookup per fragment) and with every node marked (0x40000000: the table is
read, every walker looks up itself) -- on synthetic spectra and on spectra whose peaks sit AT the window ends of the
fragments of random site assignments (+- 0..40 float32 ulps, +- a quarter of the tolerance)."""
import os

import numpy as np
import pytest

import switches
from conftest import checker_kind
from oracle import harness, orc
from pyascore_amd import synth

pytestmark = pytest.mark.gpu


def _edge_spectra(batch, settings, rng, frac_ulp):
    res = synth.RESIDUE_MASS
    err = settings["mz_error"]
    mod = np.float32(settings["mod_mass"])
    offs_of = {"b": 0.0, "c": 17.026549, "y": 18.010565, "z": 18.010565 - 17.026549}
    mzs, its, offs = [], [], [0]
    for i in range(batch["n_psm"]):
        pep = bytes(batch["pep"][batch["pep_off"][i]:batch["pep_off"][i + 1]]).decode()
        sites = [p for p, ch in enumerate(pep) if ch in settings["mod_group"]]
        ions = []
        for _ in range(8):
            md = set(rng.choice(sites, size=int(batch["n_of_mod"][i]), replace=False).tolist())
            for t in settings["fragment_types"]:
                order = range(len(pep) - 1) if t in "bc" else range(len(pep) - 1, 0, -1)
                run = np.float32(0.0)
                for p in order:
                    r = np.float32(res[pep[p]])
                    if p in md:
                        r = np.float32(r + mod)
                    run = np.float32(r + run)
                    ions.append(float(np.float32(float(run) + offs_of[t] + 1.007825)))
        ions = np.asarray(ions)
        side = rng.choice([-1.0, 1.0], ions.size)
        ulp = np.spacing(ions.astype(np.float32)).astype(np.float64)
        off = np.where(rng.random(ions.size) < frac_ulp, ulp * rng.integers(-40, 41, ions.size), rng.uniform(-err / 4, err / 4, ions.size))
        own = batch["mz"][batch["peak_off"][i]:batch["peak_off"][i + 1]][::3]
        m = np.concatenate([ions + side * err + off, own])
        m = np.sort(m[m > 50.0])
        mzs.append(m)
        its.append(rng.lognormal(5.0, 1.0, m.size))
        offs.append(offs[-1] + m.size)
    return dict(batch, mz=np.concatenate(mzs), intensity=np.concatenate(its), peak_off=np.asarray(offs, np.int64))


# thousands of site assignments (score_big.hip: one PSM per 8-wavefront workgroup) ...
CASES = [({}, {}), (dict(L=24, n_sites=14, n_mod=6), {}), (dict(L=40, n_sites=12, n_mod=4), dict(mz_error=0.2)),
         (dict(L=18, n_sites=13, n_mod=7), dict(mz_error=0.01)), ({}, dict(fragment_types="cz")),
         (dict(L=35, n_sites=16, n_mod=3), dict(fragment_types="zb", mz_error=0.45)),
         # ... and 65 to 1024 (score_cnt.hip: one PSM per wavefront): C(12,4) = 495, C(9,4) = 126, C(11,2) = 55 < 65 (walkers), C(13,2) = 78,
         # a 64-residue peptide, the N-terminus and a lysine as sites
         (dict(L=40, n_sites=12, n_mod=4), {}), (dict(L=12, n_sites=9, n_mod=4), dict(mz_error=0.3)), (dict(L=30, n_sites=11, n_mod=2), {}),
         (dict(L=20, n_sites=13, n_mod=2), dict(fragment_types="zc", mz_error=0.02)), (dict(L=64, n_sites=10, n_mod=5), dict(mz_error=0.1))]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_records_with_the_table_without_it_and_with_every_node_marked(case, monkeypatch):
    from pyascore_amd import PyAscore
    over, st_over = CASES[case]
    monkeypatch.setenv("PYA_NO_TINY", "1")
    monkeypatch.setenv("PYA_PLAIN_MIN", "0")
    batch, settings = synth.make_batch("cfg5", n_psm=6, seed=40 + case, **over)
    settings = dict(settings, **st_over)
    rng = np.random.default_rng(case)
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind())
    for label, b2 in (("plain", batch), ("edges", _edge_spectra(batch, settings, rng, 0.5)), ("edges_wide", _edge_spectra(batch, settings, rng, 0.0))):
        want = chk.score_batch(b2, int(b2["n_of_mod"].max()))
        recs = {}
        for mode, dbg in (("table", None), ("no_table", str(0x8000)), ("all_marked", str(0x40000000))):
            if dbg is None:
                monkeypatch.delenv("PYA_DEBUG", raising=False)
            else:
                monkeypatch.setenv("PYA_DEBUG", dbg)
            gpu = harness.make_scorer(PyAscore, settings)
            got = gpu.score_batch(b2)
            for key in want:
                assert np.array_equal(got[key], want[key]), (label, mode, key)
            gpu.score_batch(b2, keep=True)
            recs[mode] = gpu.batch_pep_scores()
        for mode in ("table", "all_marked"):
            for key in ("rec_off", "sig_bits", "counts", "weighted_score", "total_fragments"):
                assert np.array_equal(recs[mode][key], recs["no_table"][key]), (label, mode, key)
        # ... and every record of the first PSM against the reference's own
        chk.score(**synth.unpack_psm(b2, 0))
        raw = chk.raw_pep_scores()
        a, e = recs["table"]["rec_off"][0], recs["table"]["rec_off"][1]
        bits = (raw["signature"].astype(np.uint64) << np.arange(raw["signature"].shape[1], dtype=np.uint64)).sum(axis=1)
        assert np.array_equal(recs["table"]["sig_bits"][a:e], bits.astype(np.uint64)), label
        assert np.array_equal(recs["table"]["counts"][a:e], raw["counts"]) and np.array_equal(recs["table"]["weighted_score"][a:e], raw["weighted_score"]), label


# General settings (score_cntg.hip): neutral losses, charges, several ion types per direction.  The count nodes apply when
# every modifiable residue has the same pair of loss classes; the kernel walks the rest.
GENERAL = [
    ("cfg4", {}, {}),                                                                                # b/y/c/z, charge <= 4, loss on modified s/t/y
    ("cfg2", dict(fragment_types="yb", max_charge=2), {}),                                           # charges only
    ("cfg2", dict(fragment_types="Zc", max_charge=3, neutral_loss=("STY", 18.01528)), {}),           # loss on the UNMODIFIED sites
    ("cfg3", dict(max_charge=2, neutral_loss=("sty", 97.9769)), dict(mz_error=0.3)),                  # up to 495 site assignments
    ("cfg2", dict(max_charge=2, neutral_loss=("st", 97.9769)), {}),                                   # Y sites without the loss: classes differ -> walked
    ("cfg2", dict(fragment_types="bycz", max_charge=2, neutral_loss=("sty", 97.9769)), dict(mod_group="nSTY")),   # the N-terminus as a site
    ("cfg4", {}, dict(neutral_losses=[["sty", 97.9769], ["ST", 18.01528], ["m", 63.998]])),          # three loss masses
    ("cfg2", dict(L=30, n_sites=10, n_mod=4, max_charge=3, neutral_loss=("sty", 97.9769)), dict(mz_error=0.01)),
]


@pytest.mark.parametrize("case", range(len(GENERAL)))
def test_general_settings_records(case, monkeypatch):
    from pyascore_amd import PyAscore
    cfg, over, st_over = GENERAL[case]
    monkeypatch.setenv("PYA_NO_TINY", "1")
    monkeypatch.setenv("PYA_PLAIN_MIN", "0")
    batch, settings = synth.make_batch(cfg, n_psm=24, seed=70 + case, **over)
    settings = dict(settings, **st_over)
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=checker_kind())
    want = chk.score_batch(batch, int(batch["n_of_mod"].max()))
    recs = {}
    for mode, dbg in (("table", None), ("no_table", str(0x8000)), ("all_walked", str(0x40000000))):
        if dbg is None:
            monkeypatch.delenv("PYA_DEBUG", raising=False)
        else:
            monkeypatch.setenv("PYA_DEBUG", dbg)
        gpu = harness.make_scorer(PyAscore, settings)
        got = gpu.score_batch(batch)
        for key in want:
            assert np.array_equal(got[key], want[key]), (mode, key)
        gpu.score_batch(batch, keep=True)
        recs[mode] = gpu.batch_pep_scores()
    for mode in ("table", "all_walked"):
        for key in ("rec_off", "sig_bits", "counts", "weighted_score", "total_fragments"):
            assert np.array_equal(recs[mode][key], recs["no_table"][key]), (mode, key)
    for i in range(3):
        chk.score(**synth.unpack_psm(batch, i))
        raw = chk.raw_pep_scores()
        a, e = recs["table"]["rec_off"][i], recs["table"]["rec_off"][i + 1]
        assert np.array_equal(recs["table"]["counts"][a:e], raw["counts"]) and np.array_equal(recs["table"]["weighted_score"][a:e], raw["weighted_score"]), i
        assert np.array_equal(recs["table"]["total_fragments"][a:e], raw["total_fragments"]), i
