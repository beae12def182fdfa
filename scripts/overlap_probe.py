"""What would pipelining the scoring and the localize kernels of a batch buy?  (through gpurun)
    python scripts/overlap_probe.py cfg4 [chunks]
The batch as ONE plan on one stream against the same batch cut into `chunks` plans that run alternately on two streams
(stream 0: chunks 0, 2, ...; stream 1: chunks 1, 3, ...; stream 1 starts a chunk's scoring late): a chunk's localize kernel
then runs beside the next chunk's scoring kernel.  Wall time per pass over the batch, 20 passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import harness
from pyascore_amd import PyAscore, synth
from pyascore_amd.device import DevicePlan

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = {"cfg3": 125000}.get(cfg)
desc = synth.describe(cfg, n_psm=n, seed=1000) if n else synth.describe(cfg, seed=1000)
batch = synth.make_slice(desc)
dev = torch.device("cuda", 0)
s = harness.make_scorer(PyAscore, desc["settings"])


def timed(fn, passes=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(passes):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / passes


mz, it = torch.from_numpy(batch["mz"]).to(dev), torch.from_numpy(batch["intensity"]).to(dev)
one = DevicePlan(s, batch)
t_one = timed(lambda: one.run(mz, it))
one.check()
one.close()
N = batch["n_psm"]
cuts = [N * c // chunks for c in range(chunks + 1)]
parts = [synth.slice_batch(batch, cuts[c], cuts[c + 1]) for c in range(chunks)]
plans = [DevicePlan(s, p) for p in parts]
tens = [(torch.from_numpy(np.ascontiguousarray(p["mz"])).to(dev), torch.from_numpy(np.ascontiguousarray(p["intensity"])).to(dev)) for p in parts]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def split_pass():
    for c, pl in enumerate(plans):
        with torch.cuda.stream(streams[c & 1]):
            pl.run(*tens[c])


def serial_pass():
    for c, pl in enumerate(plans):
        pl.run(*tens[c])


t_serial = timed(serial_pass)
t_split = timed(split_pass)
for pl in plans:
    pl.check()
    pl.close()
print("%s: one plan %.3f ms | %d plans one stream %.3f ms | %d plans on two streams %.3f ms (%.1f %% of one plan)"
      % (cfg, t_one, chunks, t_serial, chunks, t_split, 100.0 * t_split / t_one))
