"""Raw pya_score_one latency through ctypes, without the PyAscore wrapper (through gpurun): python scripts/raw_probe.py"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import harness
from pyascore_amd import PyAscore, synth, _lib
batch, settings = synth.make_batch("cfg2", n_psm=200, seed=5)
gpu = harness.make_scorer(PyAscore, settings)
kw = synth.unpack_psm(batch, 3)
gpu.score(**kw)
one = gpu._one
mz, it = kw["mz_arr"], kw["int_arr"]
pep = np.frombuffer(kw["peptide"].encode(), dtype=np.uint8)
ap = np.zeros(0, np.uint32); am = np.zeros(0, np.float32)
lib = gpu._lib
n = 5000
t = time.perf_counter()
for i in range(n):
    lib.pya_score_one(gpu._h, mz.ctypes.data, it.ctypes.data, mz.size, pep.ctypes.data, pep.size, 3, 1, ap.ctypes.data, am.ctypes.data, 0, 0, C.byref(one["results"]))
dt = time.perf_counter() - t
print("raw pya_score_one via ctypes: %.1f us" % (1e6 * dt / n))
args = (gpu._h, mz.ctypes.data, it.ctypes.data, mz.size, pep.ctypes.data, pep.size, 3, 1, ap.ctypes.data, am.ctypes.data, 0, 0, C.byref(one["results"]))
f = lib.pya_score_one
t = time.perf_counter()
for i in range(n):
    f(*args)
dt = time.perf_counter() - t
print("raw, prebuilt args: %.1f us" % (1e6 * dt / n))
os.environ["PYA_HOST_TIMING"] = "1"
