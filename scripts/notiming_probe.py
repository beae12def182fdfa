#!/usr/bin/env python
"""What the per-family timing events of a plan cost a step (through gpurun): the same steps with and without them.
    python scripts/notiming_probe.py cfg2 [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
desc = synth.describe(cfg, seed=1000) if cfg != "cfg3" else synth.describe(cfg, 125000, seed=1000)
batch = synth.make_slice(desc)
s = harness.make_scorer(PyAscore, desc["settings"])
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
for timing in (True, False, True, False):
    plan = DevicePlan(s, batch, timing=timing)
    for _ in range(5):
        plan.run(mz, it)
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(steps):
            plan.run(mz, it)
        torch.cuda.synchronize()
        best.append(1e3 * (time.perf_counter() - t) / steps)
    best.sort()
    print("%s timing events %-5s: median %.4f ms/step (min %.4f)" % (cfg, timing, best[2], best[0]))
    plan.check()
    plan.close()
