#!/usr/bin/env python
"""One-off soak of the general kernel on a GPU box: python scripts/soak_general.py <first seed> <last seed>
Random PSMs beyond a limit of the fast kernels -- peptides of 65-255 residues, or 16-20 sites (up to C(20,6) = 38 760
site assignments), or n_top 11-16 -- under random scorer settings (charges 1-3, tolerance 0.02-0.5, b / y / by / yb / cz /
bycz, neutral loss on or off, fixed modifications), each PSM through PyAscore.score() bit for bit against the reference's
C++ core: best sequence, PepScore, Ascores, alternative sites, and the retained records in bulk."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
bad_total = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 3))
    n_top = 10
    if kind == 0:
        L, ns, k = int(rng.integers(65, 256)), int(rng.integers(2, 9)), int(rng.integers(1, 4))
    elif kind == 1:
        L, ns, k = int(rng.integers(30, 60)), int(rng.integers(16, 21)), int(rng.integers(4, 7))
    else:
        L, ns, k, n_top = int(rng.integers(8, 40)), int(rng.integers(2, 8)), int(rng.integers(1, 4)), int(rng.integers(11, 17))
    k = min(k, ns - 1) if rng.random() < 0.9 else ns
    over = dict(L=L, n_sites=ns, n_mod=max(k, 1))
    if rng.random() < 0.5: over["max_charge"] = int(rng.integers(1, 4))
    if rng.random() < 0.5: over["mz_error"] = float(rng.choice([0.02, 0.05, 0.3, 0.5]))
    if rng.random() < 0.5: over["fragment_types"] = str(rng.choice(["b", "y", "by", "yb", "cz", "bycz"]))
    if rng.random() < 0.3 and kind != 1: over["neutral_loss"] = ("sty", 97.9769)
    # (the score table covers 4 096 theoretical fragments per site assignment)
    types = len(over.get("fragment_types", "by"))
    if (L - 1) * over.get("max_charge", 1) * (3 if "neutral_loss" in over else 1) * types > 4096:
        over["max_charge"] = 1
        over.pop("neutral_loss", None)
    batch, st = synth.make_batch("cfg2", n_psm=2, seed=seed, **over)
    st = dict(st, n_top=n_top)
    gpu = harness.make_scorer(PyAscore, st)
    chk = harness.make_scorer(orc.OracleAscore, st, kind="ref")
    bad = 0
    for i in range(batch["n_psm"]):
        kw = synth.unpack_psm(batch, i)
        if rng.random() < 0.3:
            kw["aux_mod_pos"] = np.array([0, int(rng.integers(1, L + 1))], np.uint32)
            kw["aux_mod_mass"] = np.array([42.010565, 15.9949], np.float32)
        gpu.score(**kw)
        chk.score(**kw)
        ok = (gpu.best_sequence == chk.best_sequence and np.float32(gpu.best_score) == np.float32(chk.best_score)
              and np.array_equal(gpu.ascores, chk.ascores) and len(gpu.alt_sites) == len(chk.alt_sites)
              and all(np.array_equal(a, b) for a, b in zip(gpu.alt_sites, chk.alt_sites)))
        raw = chk.raw_pep_scores()
        if ok and raw["weighted_score"].size and gpu._last["n_sig"] > 0:
            gpu._ensure_kept()
            got = gpu.batch_pep_scores()
            bits = (raw["signature"].astype(np.uint64) << np.arange(raw["signature"].shape[1], dtype=np.uint64)).sum(axis=1).astype(np.uint64)
            ok = np.array_equal(got["sig_bits"], bits) and np.array_equal(got["counts"], raw["counts"]) and \
                np.array_equal(got["weighted_score"], raw["weighted_score"]) and np.array_equal(got["scores"], raw["scores"])
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "psm", i, over, "n_top", n_top, flush=True)
    bad_total += bad
    if seed % 20 == 0:
        print("seed", seed, over, "n_top", n_top, "mismatches so far", bad_total, flush=True)
print("general-kernel soak done, mismatches", bad_total)
