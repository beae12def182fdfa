"""The checker over a batch on several host cores -- TEST INFRASTRUCTURE ONLY.

    score_batch_parallel(settings, batch, max_k, kind="ref", workers=None) -> the dict OracleAscore.score_batch returns

The reference's core is single-threaded and, under general settings (losses, four ion types, four charges), scores a
PSM of a dense spectrum in ~0.1 s: a few thousand PSMs are minutes on one core.  The batch is written once to /dev/shm,
contiguous slices balanced by C(sites, mods) x (L - 1) go to FRESH child processes (never forks of a process that
holds the GPU), every child writes its slice of the result arrays, the parent puts them together.

    python oracle/par_check.py <batch.npz> <lo> <hi> <max_k> <kind> <out.npz>        (the child)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def score_batch_parallel(settings, batch, max_k, kind="ref", workers=None):
    from math import comb
    n = int(batch["n_psm"])
    workers = max(1, min(workers or _cores(), n))
    L = np.diff(batch["pep_off"]).astype(np.int64)
    group = settings["mod_group"]
    letters = np.frombuffer(group.encode(), dtype=np.uint8)
    is_site = np.isin(batch["pep"], letters)
    n_sites = np.add.reduceat(is_site.astype(np.int64), batch["pep_off"][:-1].astype(np.int64)) if n else np.zeros(0, np.int64)
    n_sites = np.where(L > 0, n_sites, 0)
    work = np.array([comb(int(s), int(k)) if 0 <= k <= s else 1 for s, k in zip(n_sites, batch["n_of_mod"])], np.float64)
    work = (work + 4.0) * np.maximum(L - 1, 1) * batch["max_charge"]
    cum = np.concatenate([[0.0], np.cumsum(work)])
    cuts = sorted(set(int(np.searchsorted(cum, cum[-1] * i / workers)) for i in range(workers + 1)) | {0, n})
    slices = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    tmp = tempfile.mkdtemp(prefix="pya_chk_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(tmp, "batch.npz")
    try:
        np.savez(path, settings=np.asarray(json.dumps(settings)), **{k: np.asarray(v) for k, v in batch.items()})
        outs = [os.path.join(tmp, "out%d.npz" % i) for i in range(len(slices))]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), path, str(lo), str(hi), str(max_k), kind, out])
                 for (lo, hi), out in zip(slices, outs)]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise RuntimeError("checker worker failed: exit codes %s" % rcs)
        parts = [np.load(o) for o in outs]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0].files}
    finally:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


def _child():
    path, lo, hi, max_k, kind, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    from oracle import harness, orc
    from pyascore_amd.synth import slice_batch
    z = np.load(path, allow_pickle=False)
    settings = json.loads(str(z["settings"]))
    batch = {k: z[k] for k in z.files if k != "settings"}
    batch["n_psm"] = int(batch["n_psm"])
    part = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in slice_batch(batch, lo, hi).items()}
    res = harness.make_scorer(orc.OracleAscore, settings, kind=kind).score_batch(part, max_k)
    np.savez(out, **res)


if __name__ == "__main__":
    _child()
