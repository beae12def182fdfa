"""Parity probe of the selection route of the binning (bin_select.hip.h) on random dense spectra (through gpurun):
    python scripts/dense_check.py [first seed] [last seed]
Per seed: 48 PSMs of cfg2's shape with a random noise count (700 .. 7 500 peaks), random window width, intensities of a
random kind (lognormal, integer counts, narrow range, quantised to a few levels, huge dynamic range with zeros), with /
without isotope satellites -- through the default route (selection for the dense classes) and with selection forced
(PYA_BIN_SELECT_MIN=0, small survivor room on every third seed), against the reference's C++ core, bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import harness, par_check, orc
from pyascore_amd import PyAscore, synth
import switches

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kind = "ref" if orc.available("ref") else "oracle"
bad_total = 0
for seed in range(first, last):
    rng = np.random.default_rng([seed, 0xDE5E])
    n_noise = int(rng.integers(700, 7500))
    desc = synth.describe("cfg2", n_psm=48, seed=9000 + seed, n_noise=n_noise, isotopes=bool(rng.integers(0, 2)))
    batch = synth.make_slice(desc)
    settings = dict(desc["settings"], bin_size=float(rng.choice([100.0, 100.0, 50.0, 250.0, 37.5])))
    it = batch["intensity"]
    how = int(rng.integers(0, 5))
    if how == 1:
        it = np.floor(it / np.median(it) * float(rng.choice([5.0, 40.0, 400.0]))) + 1.0
    elif how == 2:
        it = 1000.0 + (it % 1.0) * float(rng.choice([3.0, 300.0, 3000.0]))
    elif how == 3:
        it = np.exp2(np.floor(np.log2(it) * 2.0) / 2.0)          # half-octave levels: whole buckets of equal keys
    elif how == 4:
        it = it * np.exp(rng.normal(0.0, 12.0, it.size))          # 2^±50: beyond the keys' dynamic range
        it[rng.random(it.size) < 0.02] = 0.0
    batch = dict(batch, intensity=np.ascontiguousarray(it))
    k = int(batch["n_of_mod"].max())
    want = par_check.score_batch_parallel(settings, batch, k, kind=kind)
    gpu = harness.make_scorer(PyAscore, settings)
    for route in ("default", "forced"):
        for name in ("PYA_BIN_SELECT_MIN", "PYA_BIN_SELECT_SCAP"):
            os.environ.pop(name, None)
        if route == "forced":
            os.environ["PYA_BIN_SELECT_MIN"] = "0"
            if seed % 3 == 0:
                os.environ["PYA_BIN_SELECT_SCAP"] = "256"
        switches.from_env(gpu)
        got = gpu.score_batch(batch)
        for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
            bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
            if bad.size:
                bad_total += bad.size
                print("seed %d (%d noise peaks, intensities %d, bin %g) %s: %s differs for PSMs %s" % (seed, n_noise, how, settings["bin_size"], route, key, bad[:8]), flush=True)
print("dense_check seeds [%d, %d) MISMATCHES %d" % (first, last, bad_total))
