"""Quick parity probe of the hash route of the general localize kernel (through gpurun):
    python scripts/hash_check.py [n] [seed]
Every case runs under: the default (hash route), PYA_DEBUG=16384 (every in-span ion through the exact run walk),
PYA_DEBUG=0x20000000 (no closed forms for an ion with one neighbour),
PYA_DEBUG=8192 (everything declined -> list-based kernel)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYA_NO_TINY", "1")
os.environ.setdefault("PYA_PLAIN_MIN", "0")
os.environ["PYA_NO_PLAIN"] = "1"
os.environ["PYA_NO_FUSED"] = "1"
from oracle import harness, orc
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31
cases = [("cfg4", n, {}),
         ("cfg2", n, dict(fragment_types="yb", max_charge=2)),
         ("cfg2", n, dict(fragment_types="Zc", max_charge=3, neutral_loss=("STY", 18.01528))),
         ("cfg3", 2 * n, dict(mz_error=0.3, max_charge=2)),
         ("cfg2", n, dict(mz_error=0.5)),
         ("cfg4", n, dict(mod_mass=57.02146)),          # glycine: in-span ions land on other prefixes' ions
         ("cfg4", n, dict(mz_error=0.2)),
         ("cfg4", n, dict(mz_error=0.01)),              # tau and the rounding allowance are of the same order
         ("cfg4", n, dict(mz_error=0.007)),
         ("cfg4", n, dict(mz_error=0.005)),             # ... and here the route declines (allowance > tolerance / 2)
         ("cfg4", n, dict(mz_error=1.5)),               # everything has neighbours
         ("cfg3", 2 * n, {})]
bad_total = 0
for cfg, nn, over in cases:
    over = dict(over)
    mod_mass = over.pop("mod_mass", None)
    batch, settings = synth.make_batch(cfg, n_psm=nn, seed=seed, **over)
    if mod_mass is not None:
        settings["mod_mass"] = mod_mass
        over["mod_mass"] = mod_mass
    kind = "ref" if orc.available("ref") else "oracle"
    want = None
    for mode, dbg in (("hash", None), ("exact", "16384"), ("noclosed", str(0x20000000)), ("declined", "8192"), ("walkers", "NO_NODES"), ("fewnodes", "NODE_CAP")):
        for v in ("PYA_DEBUG", "PYA_NO_NODES", "PYA_NODE_CAP"):
            os.environ.pop(v, None)
        if dbg == "NO_NODES":
            os.environ["PYA_NO_NODES"] = "1"
        elif dbg == "NODE_CAP":
            os.environ["PYA_NODE_CAP"] = "150"
        elif dbg is not None:
            os.environ["PYA_DEBUG"] = dbg
        gpu = harness.make_scorer(PyAscore, settings)
        t = time.time(); got = gpu.score_batch(batch); dt = time.time() - t
        if want is None:
            want = harness.make_scorer(orc.OracleAscore, settings, kind=kind).score_batch(batch, got["ascores"].shape[1])
        nbad = 0
        for key in ("n_sig", "best_sig", "best_score", "alt_mask", "ascores"):
            bad = np.flatnonzero(np.any(np.atleast_2d((got[key] != want[key]).T), axis=0))
            if bad.size:
                nbad += bad.size
                print("%s %s [%s]: %s differs for %d PSMs, first %s: got %s want %s" % (cfg, over, mode, key, bad.size, bad[:5], got[key][bad[0]], want[key][bad[0]]), flush=True)
        bad_total += nbad
        print("%s %s [%s] n=%d checked (%.3f s) bad=%d" % (cfg, over, mode, nn, dt, nbad), flush=True)
print("MISMATCHES", bad_total)
