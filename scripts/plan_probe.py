import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["PYA_HOST_TIMING"] = "1"
import numpy as np, torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan
for cfg, n in (("cfg2", None), ("cfg3", 125000), ("cfg4", None)):
    desc = synth.describe(cfg, n_psm=n, seed=1000) if n else synth.describe(cfg, seed=1000)
    batch = synth.make_slice(desc)
    s = harness.make_scorer(PyAscore, desc["settings"])
    DevicePlan(s, batch).close()
    print("==", cfg, "second plan:", flush=True); sys.stderr.flush()
    t = time.perf_counter(); p = DevicePlan(s, batch); dt = time.perf_counter() - t
    print("== %s plan_ms %.2f" % (cfg, 1e3 * dt), flush=True)
    p.close()
