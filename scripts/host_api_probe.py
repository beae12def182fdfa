import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pyascore_amd import PyAscore, synth
from oracle import harness
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
desc = synth.describe(cfg, seed=1000)
batch = synth.make_slice(desc)
s = harness.make_scorer(PyAscore, desc["settings"])
s.score_batch(batch)
for rep in range(3):
    t = time.perf_counter(); s.score_batch(batch); dt = time.perf_counter() - t
    print("host_api %s: %.2f ms  %.2f M PSMs/s" % (cfg, dt * 1e3, batch["n_psm"] / dt / 1e6), flush=True)
