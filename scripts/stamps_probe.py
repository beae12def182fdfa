#!/usr/bin/env python
"""Phase shares of the kernels of one device-resident plan (diagnostic build, PYA_STAMPS=1):
    PYA_STAMPS=1 python scripts/stamps_probe.py cfg2 [max_charge]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
over = dict(max_charge=int(sys.argv[2])) if len(sys.argv) > 2 else {}
desc = synth.describe(cfg, seed=1000, **over)
if cfg == "cfg3":
    desc = synth.describe(cfg, 125000, seed=1000, **over)
batch = synth.make_slice(desc)
s = harness.make_scorer(PyAscore, desc["settings"])
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
plan = DevicePlan(s, batch)
for _ in range(3):
    plan.run(mz, it)
plan.check()
plan.close()
