"""Reproducer hunt for the empty result of PyAscore.score() (tests/test_gpu_handover_stress.py failed once at the FIRST
score() of a PSM with 3003 site assignments: best_sequence '' = n_sig <= 0 with an OK status).

python scripts/handover_repro.py [seconds] [fresh|same] [load|noload]
  fresh: a new scorer (new library handle) per trial, a few small PSMs first, then the heavy one;
  same:  one scorer, the heavy PSM between runs of small ones.
Prints every deviating result with everything the last call left, then a count."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyascore_amd import PyAscore, synth  # noqa: E402

_LOAD = r"""
import os, sys
sys.path.insert(0, %r)
from pyascore_amd import PyAscore, synth
batch, st = synth.make_batch("cfg2", n_psm=20000, seed=5)
s = PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], st["mz_error"], st["fragment_types"])
print("ready", flush=True)
n = 0
while not os.path.exists(sys.argv[1]):
    s.score_batch(batch)
    n += 1
print("batches", n, flush=True)
"""


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.
    mode = sys.argv[2] if len(sys.argv) > 2 else "fresh"
    with_load = (sys.argv[3] if len(sys.argv) > 3 else "load") == "load"
    small_b, st = synth.make_batch("cfg3", n_psm=60, seed=9003)
    small = [synth.unpack_psm(small_b, i) for i in range(60)]
    big_b, _ = synth.make_batch("cfg5", n_psm=4, seed=9100)
    big = [synth.unpack_psm(big_b, i) for i in range(4)]
    st = dict(st, mz_error=0.05)
    mk = lambda: PyAscore(st["bin_size"], st["n_top"], st["mod_group"], st["mod_mass"], st["mz_error"], st["fragment_types"])  # noqa: E731
    stop = "/tmp/handover_repro_stop_%d" % os.getpid()
    load = None
    if with_load:
        load = subprocess.Popen([sys.executable, "-c", _LOAD % ROOT, stop], stdout=subprocess.PIPE, text=True)
        assert load.stdout.readline().strip() == "ready"
    rng = np.random.default_rng(7)
    truth = {}
    bad = trials = 0
    gpu = mk()
    for j, kw in enumerate(big):                              # the answers, taken three times each and required equal
        got = []
        for _ in range(3):
            gpu.score(**kw)
            got.append((gpu.best_sequence, float(gpu.best_score), tuple(np.asarray(gpu.ascores).tolist()), gpu._last["n_sig"]))
        assert got[0] == got[1] == got[2] and got[0][3] > 0, got
        truth[j] = got[0]
    t0 = time.time()
    try:
        while time.time() - t0 < seconds:
            if mode == "fresh":
                gpu = mk()
            for _ in range(int(rng.integers(0, 6))):
                kw = small[int(rng.integers(len(small)))]
                gpu.score(**kw)
                what = rng.random()
                if what < 0.4:
                    _ = gpu.pep_scores
                elif what < 0.6:
                    _ = gpu.alt_sites
            j = int(rng.integers(4))
            gpu.score(**big[j])
            trials += 1
            last = gpu._last
            got = (gpu.best_sequence, float(gpu.best_score), tuple(np.asarray(gpu.ascores).tolist()), last["n_sig"])
            if got != truth[j]:
                bad += 1
                print("DEVIATION trial %d big %d: n_sig %r best_score %r best_sig %#x lazy %r ascores %r alt %r batch_n %r" % (
                    trials, j, last["n_sig"], last["best_score"], int(last["best_sig"]), last.get("lazy"),
                    np.asarray(last["ascores"]).tolist(), np.asarray(last["alt_mask"]).tolist(), gpu._batch_n), flush=True)
                n_rec = len(gpu.pep_scores)
                gpu.score(**big[j])
                again = (gpu.best_sequence, float(gpu.best_score), tuple(np.asarray(gpu.ascores).tolist()), gpu._last["n_sig"])
                print("   records now %d; scored again: %s" % (n_rec, "as expected" if again == truth[j] else repr(again)), flush=True)
    finally:
        if load is not None:
            open(stop, "w").close()
            print(load.communicate(timeout=300)[0].strip())
            os.remove(stop)
    print("%s %s: %d deviations in %d trials, %.0f s" % (mode, "load" if with_load else "noload", bad, trials, time.time() - t0))


if __name__ == "__main__":
    main()
