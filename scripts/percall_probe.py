"""Per-call latency of PyAscore.score (through gpurun): python scripts/percall_probe.py [n_calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, orc
from pyascore_amd import PyAscore, synth

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for cfg in ("cfg1", "cfg2"):
    batch, settings = synth.make_batch(cfg, n_psm=200, seed=5)
    gpu = harness.make_scorer(PyAscore, settings)
    psms = [synth.unpack_psm(batch, i) for i in range(batch["n_psm"])]
    for kw in psms[:20]:
        gpu.score(**kw)
    t = time.perf_counter()
    for i in range(n_calls):
        gpu.score(**psms[i % len(psms)])
        _ = gpu.best_score, gpu.ascores
    dt = time.perf_counter() - t
    print("%s: score() %.1f us per call = %.0f PSMs/s" % (cfg, 1e6 * dt / n_calls, n_calls / dt))
    t = time.perf_counter()
    for i in range(300):
        gpu.score(**psms[i % len(psms)])
        _ = gpu.pep_scores
    print("%s: score() + pep_scores %.1f us per call" % (cfg, 1e6 * (time.perf_counter() - t) / 300))
    kind = "ref" if orc.available("ref") else "oracle"
    chk = harness.make_scorer(orc.OracleAscore, settings, kind=kind)
    t = time.perf_counter()
    for i in range(n_calls):
        chk.score(**psms[i % len(psms)])
    dt = time.perf_counter() - t
    print("%s: checker (%s) score() %.1f us per call = %.0f PSMs/s" % (cfg, kind, 1e6 * dt / n_calls, n_calls / dt))
    a = harness.collect(gpu, synth.slice_batch(batch, 0, 50), synth.unpack_psm)
    b = harness.collect(chk, synth.slice_batch(batch, 0, 50), synth.unpack_psm)
    print("   parity of 50 PSMs through the public API:", harness.compare(a, b, exact_float=True) or "ok")
