"""One CPU-baseline worker process -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).

    python oracle/cpu_worker.py <batch.npz> <lo> <hi> <reps> <start_unix_time> <kind>

Loads PSMs [lo, hi) of a CSR batch saved by bench.py (np.savez of the batch arrays plus the scorer
settings as JSON), builds ONE checker-library scorer (`kind` = "ref": the reference's own C++ core,
oracle/_ref/libascore_ref.so; "oracle": this repo's CPU restatement), sleeps until the common start
time, scores its slice `reps` times and prints one JSON line with its PSM count and its start / end
wall-clock times.  One process per core: the reference's core allocates per fragment
(unordered_map / vector churn), and threads of one process serialise on the allocator's arenas --
processes are how the reference itself is run in parallel.
Never touches the GPU and imports nothing from pyascore_amd but the CSR slicing helper.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    path, lo, hi, reps, t_start, kind = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6]
    from oracle import harness, orc
    from pyascore_amd.synth import slice_batch
    z = np.load(path, allow_pickle=False)
    settings = json.loads(str(z["settings"]))
    batch = {k: z[k] for k in z.files if k != "settings"}
    batch["n_psm"] = int(batch["n_psm"])
    part = slice_batch(batch, lo, hi)
    k = max(1, int(batch["n_of_mod"].max()))
    scorer = harness.make_scorer(orc.OracleAscore, settings, kind=kind)
    scorer.score_batch(slice_batch(part, 0, min(8, part["n_psm"])), k)        # library paged in, tables warm
    now = time.time()
    if t_start > now:
        time.sleep(t_start - now)
    t0 = time.time()
    for _ in range(reps):
        scorer.score_batch(part, k)
    t1 = time.time()
    print(json.dumps({"n": part["n_psm"] * reps, "t0": t0, "t1": t1}), flush=True)


if __name__ == "__main__":
    main()
