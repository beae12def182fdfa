#!/usr/bin/env python
"""One-off soak on a GPU box: python scripts/soak_adversarial.py <first seed> <last seed>
Random settings in which moving the modification reproduces other fragments' m/z (modification as heavy
as a residue, sub-dalton residues, wide tolerances, several charges, one ion type), random small and
medium batches, four routes of the scorer each (single launch / fused + lean / no fused / general only),
bit for bit against the reference's C++ core.  8 000 seeds take about 90 s.  (tests/test_gpu_fuzz.py has
PYA_FUZZ_SEEDS=a:b for a soak over random settings of the usual kind.)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
masses = dict(G=57.02146, S=87.03203, T=101.04768, Y=163.06333, N=114.04293, A=71.03711, K=128.09496, R=156.10111)
peps = ["SGSGTGYGK", "GSGGSGGTK", "NGSGNGTGYR", "AGSTGGYGSGK", "SGGGSAGTGNK", "GGSGGSGGTGGYK", "TGSGNR", "SGTK", "SGSGTGYGKASGTGNGSGTK"]
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    mod_mass = float(rng.choice([57.02146, 114.04293, 0.984016, 79.966331, 87.03203, 30.0]))
    mz_error = float(rng.choice([0.02, 0.05, 0.3, 0.45, 0.49, 1.5, 4.0]))
    zmax = int(rng.choice([1, 1, 2, 3]))
    ftypes = str(rng.choice(["by", "b", "y", "cz", "by"]))
    settings = dict(bin_size=100.0, n_top=10, mod_group="STYN", mod_mass=mod_mass, mz_error=mz_error, fragment_types=ftypes, neutral_losses=[])
    psms = []
    for rep in range(int(rng.integers(3, 80))):
        pep = peps[int(rng.integers(len(peps)))]
        sites = [i for i, ch in enumerate(pep) if ch in "STYN"]
        k = int(rng.integers(1, min(4, len(sites))))
        chosen = set(rng.choice(sites, size=k, replace=False).tolist())
        res = [masses[ch] + (mod_mass if i in chosen else 0.0) for i, ch in enumerate(pep)]
        frag = []; run = 0.0
        for m in res[:-1]:
            run += m; frag.append(run + 1.007825); frag.append((run + 2 * 1.007825) / 2)
        run = 18.010565
        for m in res[::-1][:-1]:
            run += m; frag.append(run + 1.007825); frag.append((run + 2 * 1.007825) / 2)
        mz = np.array(frag)
        mz = np.concatenate([mz + rng.uniform(-1.2, 1.2, mz.size) * mz_error, rng.uniform(60.0, 2200.0, 30)])
        mz = np.sort(np.abs(mz) + 1.0)
        psm = dict(mz=mz, intensity=rng.lognormal(5, 1, mz.size), peptide=pep, n_of_mod=k, max_charge=zmax)
        if rng.random() < 0.2:
            psm["aux_pos"] = np.array([pep.index("G") + 1], np.uint32); psm["aux_mass"] = np.array([-56.42], np.float32)
        psms.append(psm)
    batch = synth.pack_batch(psms)
    want = None
    for env in ({}, {"PYA_NO_TINY": "1"}, {"PYA_NO_TINY": "1", "PYA_NO_FUSED": "1"}, {"PYA_NO_TINY": "1", "PYA_NO_PLAIN": "1"}):
        for k_ in ("PYA_NO_TINY", "PYA_NO_FUSED", "PYA_NO_PLAIN"): os.environ.pop(k_, None)
        os.environ.update(env); os.environ["PYA_PLAIN_MIN"] = "0"
        got = harness.make_scorer(PyAscore, settings).score_batch(batch)
        if want is None:
            want = harness.make_scorer(orc.OracleAscore, settings, kind="ref").score_batch(batch, got["ascores"].shape[1])
        for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
            a, b = got[key], want[key]
            eq = np.array_equal(a, b) or (a.dtype.kind == "f" and np.array_equal(a.view(np.uint32), b.view(np.uint32)))
            if not eq:
                bad += 1; print("MISMATCH seed", seed, env, key, settings)
print("done", sys.argv[1], sys.argv[2], "mismatches", bad)
