#!/usr/bin/env python
"""One-off soak on a GPU box: python scripts/soak_mixed.py <first seed> <last seed>
Batches of 2 500-40 000 PSMs of mixed shapes (the cfg3 generator) under random scorer settings (fragment
charge 1-3, tolerance 0.02-0.5, ion types b / y / by / cz / bycz, neutral loss on or off) through the
default route (peak classes, LDS classes, every kernel); 400 random PSMs of each batch bit for bit against
the reference's C++ core.  40 seeds take about a minute."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
bad_total = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2500, 40000))
    over = {}
    if rng.random() < 0.5: over["max_charge"] = int(rng.integers(1, 4))
    if rng.random() < 0.5: over["mz_error"] = float(rng.choice([0.02, 0.05, 0.3, 0.5]))
    if rng.random() < 0.4: over["fragment_types"] = str(rng.choice(["b", "y", "by", "cz", "bycz"]))
    if rng.random() < 0.25: over["neutral_loss"] = ("sty", 97.9769)
    desc = synth.describe("cfg3", n, seed=seed, **over)
    batch = synth.make_slice(desc)
    st = desc["settings"]
    gpu = harness.make_scorer(PyAscore, st)
    got = gpu.score_batch(batch)
    pick = np.sort(rng.choice(n, min(n, 400), replace=False))
    k = got["ascores"].shape[1]
    def work(ids):
        chk = harness.make_scorer(orc.OracleAscore, st, kind="ref")
        bad = 0
        for i in ids:
            w = chk.score_batch(synth.slice_batch(batch, int(i), int(i) + 1), k)
            for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
                if not np.array_equal(np.atleast_1d(got[key][i]).view(np.uint8), np.atleast_1d(w[key][0]).view(np.uint8)):
                    bad += 1; print("MISMATCH seed", seed, "psm", int(i), key, over)
        return bad
    with ThreadPoolExecutor(32) as ex:
        bad = sum(ex.map(work, np.array_split(pick, 32)))
    bad_total += bad
    print("seed", seed, "n", n, over, "sampled", len(pick), "mismatches", bad, flush=True)
print("mixed-shape soak done, mismatches", bad_total)
