#!/usr/bin/env python
"""The count-node table of score_big (walk_core.hip.h) against the checker, record by record:
    python scripts/cnt_check.py [n_psm] [seed]        ->  MISMATCHES 0
PSMs with thousands of site assignments (cfg5's shape and variants: other lengths, k, ion types cz, wider / tighter
tolerances); the synthetic spectra as they are AND with peaks placed at the window edges of the fragments of random site
assignments (+- 0 .. 40 float32 ulps, +- up to the tolerance / 4: inside the bands the table cannot decide).  Three
modes must agree with each other and with the reference in every record (counts, PepScores, order): the table (default),
no table (PYA_DEBUG=0x8000), every node marked (PYA_DEBUG=0x40000000: table read, every walker looks up itself)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()
os.environ.setdefault("PYA_NO_TINY", "1")
os.environ.setdefault("PYA_PLAIN_MIN", "0")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def edge_spectra(batch, settings, rng, frac_ulp=0.5):
    res = synth.RESIDUE_MASS
    err = settings["mz_error"]
    mod = np.float32(settings["mod_mass"])
    types = settings["fragment_types"]
    offs_of = {"b": 0.0, "c": 17.026549, "y": 18.010565, "z": 18.010565 - 17.026549}
    mzs, its, offs = [], [], [0]
    for i in range(batch["n_psm"]):
        pep = bytes(batch["pep"][batch["pep_off"][i]:batch["pep_off"][i + 1]]).decode()
        sites = [p for p, ch in enumerate(pep) if ch in settings["mod_group"]]
        ions = []
        for _ in range(8):
            md = set(rng.choice(sites, size=int(batch["n_of_mod"][i]), replace=False).tolist())
            for t in types:
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


def records(scorer, batch):
    scorer.score_batch(batch, keep=True)
    return scorer.batch_pep_scores()


cases = [("cfg5", {}, {}), ("cfg5", dict(L=24, n_sites=14, n_mod=6), {}), ("cfg5", dict(L=40, n_sites=12, n_mod=4), dict(mz_error=0.2)),
         ("cfg5", dict(L=18, n_sites=13, n_mod=7), dict(mz_error=0.01)), ("cfg5", {}, dict(fragment_types="cz")),
         ("cfg5", dict(L=35, n_sites=16, n_mod=3), dict(fragment_types="zb", mz_error=0.45))]
rng = np.random.default_rng(seed)
bad_total = 0
for cfg, over, st_over in cases:
    batch, settings = synth.make_batch(cfg, n_psm=n, seed=seed, **over)
    settings = dict(settings, **st_over)
    for label, b2 in (("plain", batch), ("edges", edge_spectra(batch, settings, rng)), ("edges_wide", edge_spectra(batch, settings, rng, frac_ulp=0.0))):
        chk = harness.make_scorer(orc.OracleAscore, settings, kind="ref" if orc.available("ref") else "oracle")
        want_sum = chk.score_batch(b2, int(b2["n_of_mod"].max()))
        got = {}
        for mode, dbg in (("table", None), ("no_table", str(0x8000)), ("all_marked", str(0x40000000))):
            os.environ.pop("PYA_DEBUG", None)
            if dbg:
                os.environ["PYA_DEBUG"] = dbg
            gpu = harness.make_scorer(PyAscore, settings)
            t = time.time()
            summ = gpu.score_batch(b2)
            dt = time.time() - t
            rec = records(gpu, b2)
            got[mode] = (summ, rec)
            nbad = 0
            for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
                if not np.array_equal(summ[key], want_sum[key]):
                    nbad += 1
                    print("%s %s %s [%s]: summary %s differs" % (cfg, over, label, mode, key), flush=True)
            bad_total += nbad
        base = got["no_table"][1]
        for mode in ("table", "all_marked"):
            for key in ("rec_off", "sig_bits", "counts", "weighted_score", "total_fragments"):
                if not np.array_equal(got[mode][1][key], base[key]):
                    bad_total += 1
                    d = np.flatnonzero(np.any(np.atleast_2d(got[mode][1][key] != base[key]).reshape(base[key].shape[0], -1), axis=1)) if got[mode][1][key].shape == base[key].shape else []
                    print("%s %s %s: records of [%s] differ from [no_table] in %s (%d records)" % (cfg, over, label, mode, key, len(d)), flush=True)
        # ... and the records against the reference, PSM by PSM, for a few PSMs
        for i in range(min(3, b2["n_psm"])):
            chk.score(**synth.unpack_psm(b2, i))
            raw = chk.raw_pep_scores()
            a, bq = base["rec_off"][i], base["rec_off"][i + 1]
            bits = (raw["signature"].astype(np.uint64) << np.arange(raw["signature"].shape[1], dtype=np.uint64)).sum(axis=1)
            ok = np.array_equal(got["table"][1]["sig_bits"][a:bq], bits.astype(np.uint64)) and np.array_equal(got["table"][1]["counts"][a:bq], raw["counts"]) \
                and np.array_equal(got["table"][1]["weighted_score"][a:bq], raw["weighted_score"])
            if not ok:
                bad_total += 1
                print("%s %s %s: records of PSM %d differ from the reference" % (cfg, over, label, i), flush=True)
        print("%s %s %s %s n=%d n_sig=%d checked (%.3f s)" % (cfg, over, st_over, label, b2["n_psm"], int(want_sum["n_sig"][0]), dt), flush=True)
print("MISMATCHES", bad_total)
