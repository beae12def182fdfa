#!/usr/bin/env python
"""The general kernel (csrc/general_psm.hip: PSMs beyond a limit of the fast kernels) against the reference's C++ core:
    python scripts/general_probe.py        (through gpurun; exit text `MISMATCHES 0`)
Long peptides, more than 15 000 site assignments, more than 2 048 fragments per ion type -- results and the retained
per-signature records (one bulk call on either side)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, orc
from pyascore_amd import PyAscore, synth

rng = np.random.default_rng(5)
bad = 0
for settings, shapes in [
    (dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.05, fragment_types="by", neutral_losses=[]),
     [(65, 3, 1, 1), (80, 5, 2, 2), (120, 6, 3, 1), (200, 4, 2, 1), (255, 3, 1, 1), (40, 20, 5, 1), (30, 17, 8, 1), (70, 4, 4, 1), (90, 2, 0, 1)]),
    (dict(bin_size=100.0, n_top=10, mod_group="STY", mod_mass=79.966331, mz_error=0.02, fragment_types="b", neutral_losses=[["sty", 97.9769], ["ST", 18.01528]]),
     [(60, 5, 2, 8), (100, 4, 2, 3)]),
    (dict(bin_size=100.0, n_top=10, mod_group="nKc", mod_mass=42.010565, mz_error=0.5, fragment_types="Zc", neutral_losses=[]),
     [(100, 5, 2, 2)]),
]:
    gpu = harness.make_scorer(PyAscore, settings)
    ref = harness.make_scorer(orc.OracleAscore, settings, kind="ref")
    for L, nsites, k, z in shapes:
        pep = list(rng.choice(list("ACDEFGHILMNPQRVW"), L))
        letters = [c for c in settings["mod_group"] if c.isupper()]
        for p in rng.choice(L, nsites, replace=False):
            pep[p] = rng.choice(letters)
        pep = "".join(pep)
        mz, it = np.sort(rng.uniform(100.0, 4000.0, 500)), rng.lognormal(5, 1, 500)
        aux = dict(aux_mod_pos=np.array([0, 7], np.uint32), aux_mod_mass=np.array([42.010565, 15.9949], np.float32)) if L == 80 else {}
        c = dict(mz_arr=mz, int_arr=it, peptide=pep, n_of_mod=k, max_fragment_charge=z, **aux)
        gpu.score(**c)
        ref.score(**c)
        ok = (gpu.best_sequence == ref.best_sequence and np.float32(gpu.best_score) == np.float32(ref.best_score)
              and np.array_equal(gpu.ascores, ref.ascores) and len(gpu.alt_sites) == len(ref.alt_sites)
              and all(np.array_equal(a, b) for a, b in zip(gpu.alt_sites, ref.alt_sites)))
        # the retained records, bulk on both sides
        raw = ref.raw_pep_scores()
        bits = (raw["signature"].astype(np.uint64) << np.arange(raw["signature"].shape[1], dtype=np.uint64)).sum(axis=1).astype(np.uint64) \
            if raw["signature"].size else np.zeros(0, np.uint64)
        got = gpu.batch_pep_scores() if gpu._batch_n else None
        ok2 = got is not None and np.array_equal(got["sig_bits"], bits) and np.array_equal(got["counts"], raw["counts"]) and \
            np.array_equal(got["weighted_score"], raw["weighted_score"]) and np.array_equal(got["scores"], raw["scores"]) and \
            np.array_equal(got["total_fragments"], raw["total_fragments"])
        print("L=%d sites=%d k=%d z=%d records=%d  results %s  records %s   %s %.4f %s" %
              (L, nsites, k, z, bits.size, ok, ok2, gpu.best_sequence[:30], gpu.best_score, gpu.ascores), flush=True)
        bad += (not ok) + (not ok2)
print("MISMATCHES", bad, flush=True)
