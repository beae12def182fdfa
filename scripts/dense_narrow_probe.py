"""Binning time on dense spectra whose intensities are packed into a narrow range (through gpurun):
    python scripts/dense_narrow_probe.py [n_noise] [power]
cfg2's peptides, 8 192 spectra of ~n_noise peaks with isotope satellites, intensities raised to `power` (0.25: a lognormal of
sigma 0.25-0.3 -- the whole spectrum within a few half-octave buckets).  Prints the binning kernels' time per step with the
selection route, and with the all-pairs kernel (PYA_BIN_SELECT_MIN huge) beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

n_noise = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
power = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
desc = synth.describe("cfg2", n_psm=8192, seed=1000, n_noise=n_noise, isotopes=True)
batch = synth.make_slice(desc)
batch = dict(batch, intensity=np.ascontiguousarray(batch["intensity"] ** power))
dev = torch.device("cuda", 0)
mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
for name, sw in (("selection", {}), ("all-pairs", {"PYA_BIN_SELECT_MIN": "1000000"})):
    s = harness.make_scorer(PyAscore, desc["settings"])
    for k, v in sw.items():
        s.set_debug(k, v)
    plan = DevicePlan(s, batch, timing=True)
    for _ in range(3):
        plan.run(mz, it)
    plan.check()
    steps = 10
    for _ in range(steps):
        plan.run(mz, it)
    torch.cuda.synchronize()
    ms, n = plan.timings_sum()
    print("%s: intensities ** %g, %d peaks per spectrum: binning %.3f ms per step of 8192 spectra (all kernel families: %s)"
          % (name, power, batch["peak_off"][-1] // 8192, ms[0] / max(n, 1), [round(float(x) / max(n, 1), 3) for x in ms]), flush=True)
    plan.close()
