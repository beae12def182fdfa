"""Quick parity probe of score_big's in-kernel localisation (through gpurun): python scripts/big_check.py [n] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYA_NO_TINY", "1")
os.environ.setdefault("PYA_PLAIN_MIN", "0")
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 21
bad_total = 0
for cfg, nn in (("cfg5", n), ("cfg3", 20 * n)):
    batch, settings = synth.make_batch(cfg, n_psm=nn, seed=seed)
    gpu = harness.make_scorer(PyAscore, settings)
    t = time.time(); got = gpu.score_batch(batch); dt = time.time() - t
    kind = "ref" if orc.available("ref") else "oracle"
    want = harness.make_scorer(orc.OracleAscore, settings, kind=kind).score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        if bad.size:
            bad_total += bad.size
            print("%s: %s differs for %d PSMs, first %s: got %s want %s" % (cfg, key, bad.size, bad[:5], got[key][bad[0]], want[key][bad[0]]), flush=True)
    print("%s n=%d checked (%.3f s)" % (cfg, nn, dt), flush=True)
print("MISMATCHES", bad_total)
