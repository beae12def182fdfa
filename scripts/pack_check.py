"""Quick parity probe of the packed fused kernel against the CPU checker (through gpurun):
    python scripts/pack_check.py [cfg n seed] ...   (env: PYA_PACK_G, PYA_DEBUG, PYA_PACK_PEAKS ...)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYA_PACK", "1")
os.environ.setdefault("PYA_PACK_MIN", "0")
os.environ.setdefault("PYA_PACK_GROUP_MIN", "1")
os.environ.setdefault("PYA_NO_TINY", "1")
os.environ.setdefault("PYA_PLAIN_MIN", "0")
from oracle import harness, orc          # noqa: E402
from pyascore_amd import PyAscore, synth   # noqa: E402

cases = [("cfg2", 3000, 11), ("cfg3", 6000, 12), ("cfg1", 2000, 13)]
if len(sys.argv) > 3:
    cases = [(sys.argv[i], int(sys.argv[i + 1]), int(sys.argv[i + 2])) for i in range(1, len(sys.argv) - 2, 3)]
bad_total = 0
for cfg, n, seed in cases:
    batch, settings = synth.make_batch(cfg, n_psm=n, seed=seed)
    gpu = harness.make_scorer(PyAscore, settings)
    t = time.time()
    got = gpu.score_batch(batch)
    dt = time.time() - t
    kind = "ref" if orc.available("ref") else "oracle"
    want = harness.make_scorer(orc.OracleAscore, settings, kind=kind).score_batch(batch, got["ascores"].shape[1])
    for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
        bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
        if bad.size:
            bad_total += bad.size
            print("%s seed %d: %s differs for %d PSMs, first %s: got %s want %s" % (cfg, seed, key, bad.size, bad[:5], got[key][bad[0]], want[key][bad[0]]), flush=True)
    print("%s n=%d seed=%d checked (%.3f s on the GPU path)" % (cfg, n, seed, dt), flush=True)
print("MISMATCHES", bad_total)
