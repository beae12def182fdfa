#!/usr/bin/env python
"""Cumulative time profile of the fused score+localize kernel (diagnostic build, scripts/stamps.sh):
every wave returns at stamp k (PYA_DEBUG=k<<16); the kernel's duration against k shows where a PSM's
time goes without trusting where the compiler put the s_memtime reads.
    python scripts/trunc_probe.py cfg2 "48 49 50 40 41 42 43 44 45 46 47" """
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
points = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "48 49 50 40 41 42 43 44 45 46 47").split()]
desc = synth.describe(cfg, 125000 if cfg == "cfg3" else None, seed=1000)
batch = synth.make_slice(desc)
st = desc["settings"]
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
for k in points + ["0"]:
    # "d<bits>": a raw PYA_DEBUG value (phase switches) instead of a truncation point
    os.environ["PYA_DEBUG"] = k[1:] if k.startswith("d") else str(int(k) << 16)
    s = PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], mz_error=st["mz_error"],
                 fragment_types=st["fragment_types"])
    for g, m in st.get("neutral_losses", []):
        s.add_neutral_loss(g, m)
    plan = DevicePlan(s, batch, timing=True)
    ts = []
    for _ in range(4):
        plan.run(mz, it)
        torch.cuda.synchronize()
        ts.append(plan.timings_ms())
    t = ts[-1]
    print("stop at %4s: bin %.3f score %.3f fused %.3f localize %.3f ms" % (k, t[0], t[1], t[2], t[3]), flush=True)
    plan.close()
