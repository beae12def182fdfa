#!/usr/bin/env python
"""cfg2-shaped batch at fragment charge z (what real 3+/4+ precursors look like): device-resident rate.
    python scripts/charge_probe.py 2"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

z = int(sys.argv[1]) if len(sys.argv) > 1 else 2
desc = synth.describe("cfg2", seed=1000, max_charge=z)
batch = synth.make_slice(desc)
s = harness.make_scorer(PyAscore, desc["settings"])
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
plan = DevicePlan(s, batch, timing=True)
for _ in range(5):
    plan.run(mz, it)
torch.cuda.synchronize()
t = time.perf_counter()
ms = [0.0] * 4
for _ in range(20):
    plan.run(mz, it)
    ms = [a + b for a, b in zip(ms, plan.timings_ms())]
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
plan.check()
print("cfg2 shape, fragment charge %d: %.1f M PSMs/s, %.3f ms/step, kernels %s" % (z, batch["n_psm"] / dt / 1e6, dt * 1e3, [round(m / 20, 3) for m in ms]))
