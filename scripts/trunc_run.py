#!/usr/bin/env python
"""One truncated run for a counter pass (scripts/trunc_pmc.sh): python3 scripts/trunc_run.py cfg2 <PYA_DEBUG value>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PYA_DEBUG"] = sys.argv[2] if len(sys.argv) > 2 else "0"
import torch
from pyascore_amd import PyAscore, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import switches; switches.install()   # route switches named in os.environ reach the scorers (tests/switches.py)
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
over = dict(max_charge=int(os.environ["PYA_MAX_CHARGE"])) if os.environ.get("PYA_MAX_CHARGE") else {}
desc = synth.describe(cfg, 125000 if cfg == "cfg3" else None, seed=1000, **over)
batch = synth.make_slice(desc)
st = desc["settings"]
s = PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], mz_error=st["mz_error"],
             fragment_types=st["fragment_types"])
for g, m in st.get("neutral_losses", []):
    s.add_neutral_loss(g, m)
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
plan = DevicePlan(s, batch)
for _ in range(2):
    plan.run(mz, it)
torch.cuda.synchronize()
if os.environ["PYA_DEBUG"] == "0":
    plan.check()                      # (PYA_HOST_TIMING=1: prints how many PSMs the lean kernels handed over)
plan.close()
