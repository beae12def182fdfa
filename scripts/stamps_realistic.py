#!/usr/bin/env python
"""Diagnostic build (PYA_STAMPS=1): the counters of the rare branches on the realistic-cluster data of tests/test_gpu_realistic.py
    PYA_STAMPS=1 python scripts/stamps_realistic.py general|plain [n_psm]
slots 40-44: ions with a neighbour seen by lh_resolve / closed form (a) / closed form (b) / several neighbours / left to the exact
walk; 50, 51: count nodes looked up / marked; 52-55: site assignments / with a marked node / wave rounds that walk / wave rounds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

flavour = sys.argv[1] if len(sys.argv) > 1 else "general"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
general = flavour == "general"
batch, settings = synth.make_realistic(n, seed=6001 if general else 6002, general=general, max_sites=9 if general else 12,
                                       max_mod=3 if general else 5)
s = harness.make_scorer(PyAscore, settings)
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
plan = DevicePlan(s, batch)
plan.run(mz, it)
plan.check()
plan.close()
