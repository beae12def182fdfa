#!/usr/bin/env python
"""One-off measurement: a batch whose peak counts have a long tail (99 % cfg2-like spectra of ~320
peaks, 1 % of ~4 300 peaks), scored with and without peak classes (PYA_ONE_PEAK_CLASS=1)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
from pyascore_amd.device import DevicePlan


def concat(a, b):
    out = dict(n_psm=a["n_psm"] + b["n_psm"])
    for key, off in (("mz", None), ("intensity", None), ("pep", None), ("n_of_mod", None), ("max_charge", None),
                     ("aux_pos", None), ("aux_mass", None)):
        out[key] = np.concatenate([a[key], b[key]])
    for key in ("peak_off", "pep_off", "aux_off"):
        out[key] = np.concatenate([a[key], b[key][1:] + a[key][-1]])
    return out


small, settings = synth.make_batch("cfg2", n_psm=99_000, seed=1)
rng = np.random.default_rng(2)
pep, mz, inten, counts = synth._fixed_shape(rng, 1000, 20, 6, 3, 0.05, n_noise=4000)
big = dict(n_psm=1000, mz=mz, intensity=inten, pep=pep.ravel(), pep_off=np.arange(1001, dtype=np.int64) * 20,
           peak_off=np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
           n_of_mod=np.full(1000, 3, np.int32), max_charge=np.ones(1000, np.int32),
           aux_pos=np.zeros(0, np.uint32), aux_mass=np.zeros(0, np.float32), aux_off=np.zeros(1001, np.int64))
batch = concat(small, big)
dev = torch.device("cuda", 0)
d_mz = torch.from_numpy(batch["mz"]).to(dev)
d_int = torch.from_numpy(batch["intensity"]).to(dev)
for mode in ("classes", "one"):
    if mode == "one":
        os.environ["PYA_ONE_PEAK_CLASS"] = "1"
    scorer = PyAscore(settings["bin_size"], settings["n_top"], settings["mod_group"], settings["mod_mass"],
                      settings["mz_error"], settings["fragment_types"])
    plan = DevicePlan(scorer, batch, timing=True)
    for _ in range(3):
        plan.run(d_mz, d_int)
    plan.check()
    ms = np.zeros(3)
    for _ in range(10):
        plan.run(d_mz, d_int)
        ms += np.asarray(plan.timings_ms())
    print(mode, "bin %.3f score %.3f localize %.3f ms" % tuple(ms / 10), flush=True)
    res = {k: getattr(plan, k).cpu().numpy().copy() for k in ("best_score", "best_sig", "ascores")}
    if mode == "classes":
        keep = res
    else:
        assert all(np.array_equal(keep[k], res[k]) for k in res), "results differ between class modes"
