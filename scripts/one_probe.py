"""Where the time of one PyAscore.score() call goes (through gpurun): python scripts/one_probe.py [n_calls]
   full call / the same through ctypes with prepared arguments / a ctypes call that does nothing."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness
from pyascore_amd import PyAscore, synth, _lib

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
batch, settings = synth.make_batch("cfg2", n_psm=200, seed=5)
gpu = harness.make_scorer(PyAscore, settings)
psms = [synth.unpack_psm(batch, i) for i in range(batch["n_psm"])]
for kw in psms[:50]:
    gpu.score(**kw)

def med_us(f, n):
    ts = []
    for i in range(n):
        t = time.perf_counter_ns()
        f(i)
        ts.append(time.perf_counter_ns() - t)
    ts.sort()
    return ts[len(ts) // 2] / 1e3, ts[len(ts) // 10] / 1e3, ts[9 * len(ts) // 10] / 1e3

print("PyAscore.score()            median %.1f us (p10 %.1f, p90 %.1f)" % med_us(lambda i: gpu.score(**psms[i % 200]), n_calls))
lib, h = gpu._lib, gpu._h
k = 3
res = dict(best_score=np.zeros(1, np.float32), best_sig=np.zeros(1, np.uint64), n_sig=np.zeros(1, np.int32),
           ascores=np.zeros((1, k), np.float32), alt_mask=np.zeros((1, k), np.uint64))
R = _lib.Results(k, *[v.ctypes.data_as(C.c_void_p) for v in (res["best_score"], res["best_sig"], res["n_sig"], res["ascores"], res["alt_mask"])])
prepared = []
for kw in psms:
    mz = np.ascontiguousarray(kw["mz_arr"], np.float64); it = np.ascontiguousarray(kw["int_arr"], np.float64)
    pep = np.frombuffer(kw["peptide"].encode(), np.uint8)
    prepared.append((mz, it, pep, int(kw["n_of_mod"]), int(kw.get("max_fragment_charge", 1))))
ap = np.zeros(0, np.uint32); am = np.zeros(0, np.float32)
def direct(i):
    mz, it, pep, n, z = prepared[i % 200]
    rc = lib.pya_score_one(h, mz.ctypes.data, it.ctypes.data, mz.size, pep.ctypes.data, pep.size, n, z, ap.ctypes.data, am.ctypes.data, 0, 0, C.byref(R))
    assert rc == 0
print("pya_score_one through ctypes median %.1f us (p10 %.1f, p90 %.1f)" % med_us(direct, n_calls))
ptrs = [(mz.ctypes.data, it.ctypes.data, mz.size, pep.ctypes.data, pep.size, n, z) for mz, it, pep, n, z in prepared]
apd, amd, Rb = ap.ctypes.data, am.ctypes.data, C.byref(R)
f = lib.pya_score_one
def direct2(i):
    a = ptrs[i % 200]
    f(h, a[0], a[1], a[2], a[3], a[4], a[5], a[6], apd, amd, 0, 0, Rb)
print("... with the pointers prepared too median %.1f us (p10 %.1f, p90 %.1f)" % med_us(direct2, n_calls))
print("pya_version() through ctypes   median %.2f us" % med_us(lambda i: lib.pya_version(), n_calls)[0])
ms = (C.c_double * 12)()
lib.pya_one_times(h, C.byref(ms))
for i in range(n_calls):
    direct2(i)
lib.pya_one_times(h, C.byref(ms))
print("inside pya_score_one (us, averages of %d): checks+tables %.2f, copy in %.2f, launch %.2f, wait %.2f, copy out %.2f" % ((int(ms[5]),) + tuple(ms[:5])))
print("inside the kernel (us): scalars into place %.2f, binning %.2f, scoring + localisation %.2f, rest %.2f" % tuple(ms[6:10]))
print("shader clock during the kernel: %.0f cycles / %.2f us = %.2f GHz" % (ms[10], sum(ms[6:10]), ms[10] / max(sum(ms[6:10]), 1e-9) / 1e3))
