#!/usr/bin/env python
"""Phase cycles of ONE PSM alone on the device (diagnostic build, PYA_STAMPS=1): a plan of one PSM takes the
single-launch kernel, one wavefront -- what a lone wavefront spends per phase, spectrum resident in HBM.
    PYA_STAMPS=1 python scripts/stamps_one.py [cfg] [runs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch, settings = synth.make_batch(cfg, n_psm=1, seed=5)
s = harness.make_scorer(PyAscore, settings)
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
plan = DevicePlan(s, batch)
for _ in range(runs):
    plan.run(mz, it)
    torch.cuda.synchronize()
plan.check()
print("runs", runs, "(divide the cycle sums by it)")
plan.close()
