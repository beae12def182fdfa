#!/usr/bin/env python
"""bench.py -- PSMs/s of the Ascore hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--psms M]

A *step* is one pass of the hot path (bin_spectra -> score_signatures -> rank_and_localize)
over one batch of synthetic PSMs of the named BASELINE config whose spectra are already
resident in HBM; for N > 1 every rank scores its own batch of the same size (weak scaling)
and every step ends with the single RCCL gather of the fixed-size result records to rank 0.
The job is ONE seeded description (per-PSM shapes) cut by pyascore_amd.shard.partition into
work-balanced contiguous ranges; a rank generates and scores its own range only.  --scaling strong
keeps the job size fixed as N grows (e.g. --config cfg3 --scaling strong: 1M PSMs over N GPUs).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : the dominant kernel's achieved algorithmic GB/s vs the 8 TB/s HBM peak,
                 timed with HIP events on the launch stream inside the timed region;
  cpu_baseline : the reference's own C++ core (oracle/_ref, if its prebuilt library travelled;
                 otherwise this repo's CPU restatement) timed on one host core on a bounded
                 sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "PSMs/sec (Ascore.score) at 1/2/4/8 MI355X + HBM-roofline %"


def algorithmic_bytes(batch, max_k):
    """per PSM 16*P + L + 8 + 8*n_aux + 16 + 64 (summary record)."""
    n = batch["n_psm"]
    peaks = int(batch["peak_off"][-1] - batch["peak_off"][0])
    res = int(batch["pep_off"][-1] - batch["pep_off"][0])
    aux = int(batch["aux_off"][-1] - batch["aux_off"][0])
    return 16 * peaks + res + 8 * n + 8 * aux + 16 * n + 64 * n


# the kernels between two of the plan's timing events = one family of the bench line
FAMILIES = {
    "pya_bin_spectra_kernel": ("pya_bin_spectra_kernel", "pya_bin_exact_kernel"),
    "pya_score_signatures_kernel": ("pya_score_signatures_kernel", "pya_score_big_kernel"),
    "pya_score_localize_kernel": ("pya_score_localize_kernel",),
    "pya_localize_kernel": ("pya_localize_kernel", "pya_localize_ties_kernel", "pya_localize_redo_kernel"),
}


def profiled_traffic(cfg, kernel, default_size):
    """HBM bytes per step of the kernel family `kernel` (all its instantiations and launches) from the
    newest committed PMC summary of this config (profiles/*_rocprof_<cfg>/pmc_summary.csv, collected by
    scripts/profile.sh in separate --pmc passes).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
    FETCH_SIZE tallies 128-byte requests as 64 bytes, hence the factor 2 (MI355X_MICROARCH.md, HBM
    section).  Also returns the whole path's traffic (every pya_* kernel) and the share of the
    family's SIMD cycles in which the vector ALU is occupied.  None when the run is not the profiled
    workload."""
    import csv
    import glob
    none = (None, None, None, None)
    if not default_size:
        return none
    dirs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_" + cfg)))
    if not dirs:
        return none
    path = os.path.join(dirs[-1], "pmc_summary.csv")
    fam, path_total = {}, {}
    try:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                val = float(row.get("per_step") or row["mean_value"])
                path_total[row["counter"]] = path_total.get(row["counter"], 0.0) + val
                if row["kernel"].split("<")[0] in FAMILIES.get(kernel, (kernel,)):
                    fam[row["counter"]] = fam.get(row["counter"], 0.0) + val
    except OSError:
        return none
    if "FETCH_SIZE" not in fam or "WRITE_SIZE" not in fam:
        return none
    valu = None
    if fam.get("SQ_BUSY_CYCLES") and ("SQ_ACTIVE_INST_VALU" in fam or "SQ_INSTS_VALU" in fam):
        # The bound these kernels actually run into: the vector ALUs.  SQ_ACTIVE_INST_VALU counts, in
        # quad-cycles, the time waves spend executing vector instructions (it equals SQ_INSTS_VALU here:
        # four cycles per wave64 instruction of this integer / compare / f64 mix); 1024 SIMDs;
        # SQ_BUSY_CYCLES sums 32 shader engines.  1.0 = every SIMD's VALU occupied all the time.
        quads = fam.get("SQ_ACTIVE_INST_VALU", fam.get("SQ_INSTS_VALU"))
        valu = quads * 4.0 / 1024.0 / (fam["SQ_BUSY_CYCLES"] / 32.0)
    whole = (2.0 * path_total.get("FETCH_SIZE", 0.0) + path_total.get("WRITE_SIZE", 0.0)) * 1024.0
    return (2.0 * fam["FETCH_SIZE"] + fam["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT), valu, whole


def achievable_hbm_gbs(torch, dev, nbytes=1 << 30, reps=5):
    """Read + write rate of a plain device copy of 1 GiB (the "trivial kernel" of SURVEY 8(d)): the
    HBM bandwidth a streaming kernel reaches on this box, reported beside the 8 TB/s spec."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        dst.copy_(src)
    t1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (t0.elapsed_time(t1) * 1e-3) / 1e9


def cpu_baseline(batch, settings, target_seconds=12.0):
    """The reference's C++ core (oracle/_ref; the CPU restatement if that library did not travel)
    on the host cores of this box: one scorer per thread (ctypes releases the GIL), every thread a
    contiguous slice of a bounded sample of rank 0's batch.  Reports the all-core rate as `value`
    and the single-thread rate beside it."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import harness, orc
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])
    kind = "ref" if orc.available("ref") else "oracle"
    from pyascore_amd.synth import slice_batch
    k = int(batch["n_of_mod"].max())
    cores = max(1, min(os.cpu_count() or 1, 64))
    scorers = [harness.make_scorer(orc.OracleAscore, settings, kind=kind) for _ in range(cores)]
    probe = min(200, batch["n_psm"])
    t = time.perf_counter()
    scorers[0].score_batch(slice_batch(batch, 0, probe), k)
    rate1 = probe / max(time.perf_counter() - t, 1e-9)
    # single thread: about a third of the budget; all cores: the rest
    n1 = int(min(batch["n_psm"], max(probe, rate1 * target_seconds / 3)))
    t = time.perf_counter()
    scorers[0].score_batch(slice_batch(batch, 0, n1), k)
    dt1 = time.perf_counter() - t
    # all-core rate from a short probe first (64 threads rarely scale 64x: memory-bound slices)
    probe_all = int(min(batch["n_psm"], cores * 64))
    pc = [probe_all * i // cores for i in range(cores + 1)]
    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda i: scorers[i].score_batch(slice_batch(batch, pc[i], pc[i + 1]), k) if pc[i + 1] > pc[i]
                    else None, range(cores)))
    rate_all = probe_all / max(time.perf_counter() - t, 1e-9)
    want = max(rate_all, rate1) * target_seconds * 2 / 3     # PSMs for the all-core leg
    n = int(min(batch["n_psm"], max(probe, want)))
    reps = max(1, int(round(want / n)))                      # small batches are scored several times over
    cuts = [n * i // cores for i in range(cores + 1)]

    def work(i):
        part = slice_batch(batch, cuts[i], cuts[i + 1])
        for _ in range(reps):
            scorers[i].score_batch(part, k)

    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t
    n *= reps
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for row in f:
                if row.startswith("model name"):
                    cpu_model = row.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n / dt, "unit": "PSMs/s", "cores": cores, "cpu": cpu_model,
            "kind": "reference" if kind == "ref" else "port",
            "sample": "%d PSMs (first %d of rank 0's batch x %d) in %d slices, one thread each, %.1f s"
                      % (n, n // reps, reps, cores, dt),
            "one_core": {"value": n1 / dt1, "sample": "first %d PSMs, one thread, %.1f s" % (n1, dt1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--psms", type=int, default=None, help="PSMs per GPU (default: the config's size)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU gets --psms PSMs' worth of work; strong: the job (--total) is fixed")
    ap.add_argument("--total", type=int, default=None, help="PSMs of the whole job with --scaling strong "
                    "(default: the config's size, e.g. 1M for cfg3)")
    ap.add_argument("--max-charge", type=int, default=None,
                    help="override the config's max fragment charge (real 3+/4+ precursors are scored at 2/3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the host-array legs (profiling runs: only the timed device-resident steps launch kernels)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pyascore_amd import PyAscore, shard, synth
    from pyascore_amd.device import DevicePlan
    from pyascore_amd.shard import dist_gather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PYA_BENCH_FORCE_DIST=1 runs the collective path with a world of one (the only way to exercise
    # RCCL on a single-GPU box)
    use_dist = world > 1 or bool(os.environ.get("PYA_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ONE job description (per-PSM shapes only) that every rank derives from the same seed; the job is
    # cut into contiguous ranges balanced by C(n,k) x (L-1) x types x charges (shard.partition) and
    # every rank generates the spectra of its own range only.
    n_per_gpu = args.psms or synth.CONFIGS[args.config]["n_psm"]
    if args.config == "cfg3" and args.psms is None:
        n_per_gpu = synth.CONFIGS["cfg3"]["n_psm"] // 8      # the config is quoted on 8 GPUs
    if args.scaling == "strong":
        total = args.total or (world * args.psms if args.psms else synth.CONFIGS[args.config]["n_psm"])
    else:
        total = world * n_per_gpu
    over = {} if args.max_charge is None else {"max_charge": args.max_charge}
    desc = synth.describe(args.config, n_psm=total, seed=1000, **over)
    settings = desc["settings"]
    weights = shard.work_estimate_shapes(desc["n_sites"], desc["n_mod"], desc["L"], desc["max_charge"],
                                         n_types=len(settings["fragment_types"]))
    ranges = shard.partition(weights, world)
    lo, hi = ranges[rank]
    batch = synth.make_slice(desc, lo, hi)
    job_max_k = max(1, int(desc["n_mod"].max()))
    longest = max(h - l for l, h in ranges)
    scorer = PyAscore(settings["bin_size"], settings["n_top"], settings["mod_group"], settings["mod_mass"],
                      settings["mz_error"], settings["fragment_types"], device=local_rank)
    for g, m in settings["neutral_losses"]:
        scorer.add_neutral_loss(g, m)

    d_mz = torch.from_numpy(batch["mz"]).to(dev)
    d_int = torch.from_numpy(batch["intensity"]).to(dev)
    plan = DevicePlan(scorer, batch, timing=True, max_k=job_max_k)

    in_flight = []
    # two send buffers of the job-wide record shape (longest shard x width): equal on every rank
    send = [torch.zeros((longest, shard.record_width(job_max_k)), dtype=torch.int32, device=dev) for _ in range(2)]
    flip = [0]

    def step():
        plan.run(d_mz, d_int)
        if use_dist:
            # the single RCCL gather of the path, asynchronous: the gather of this batch's packed
            # records overlaps the kernels of the next batch (at most one gather behind)
            buf = send[flip[0]]
            flip[0] ^= 1
            plan.packed_summary(out=buf[: hi - lo])
            in_flight.append(dist_gather(buf, 0, async_op=True))
            if len(in_flight) > 1:
                in_flight.pop(0)[0].wait()

    def drain():
        while in_flight:
            in_flight.pop(0)[0].wait()

    for _ in range(args.warmup):
        step()
    drain()
    plan.check()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    kern_ms = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kern_ms += np.asarray(plan.timings_ms())           # HIP events on the launch stream
    drain()                                                # every gather of the timed steps has landed
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    plan.check()
    kern_ms /= max(args.steps, 1)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        names = ["pya_bin_spectra_kernel", "pya_score_signatures_kernel", "pya_score_localize_kernel", "pya_localize_kernel"]
        dom = int(np.argmax(kern_ms))
        alg = algorithmic_bytes(batch, plan.max_k)
        achieved = alg / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
        # host API rate (host arrays in -> host results out; PCIe and host pre-pass included)
        host = None
        if not args.no_host_api:
            scorer.score_batch(batch)                          # first call allocates the workspace (reused afterwards)
            host_s = []
            for _ in range(3):
                t = time.perf_counter()
                res = scorer.score_batch(batch)
                host_s.append(time.perf_counter() - t)
            host_rate = batch["n_psm"] / min(host_s)
            # what PCIe alone costs this entry point: the same host arrays up (16 bytes per peak, pageable
            # memory as a caller holds it) and the result arrays back, nothing else
            pcie_s = []
            d_res = {k: torch.from_numpy(v).to(dev) for k, v in res.items()}
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                a = torch.from_numpy(batch["mz"]).to(dev)
                b2 = torch.from_numpy(batch["intensity"]).to(dev)
                back = [v.cpu() for v in d_res.values()]
                torch.cuda.synchronize()
                pcie_s.append(time.perf_counter() - t)
                del a, b2, back
            pcie_bytes = batch["mz"].nbytes + batch["intensity"].nbytes + sum(v.nbytes for v in res.values())
            host = {"value": host_rate, "unit": "PSMs/s", "ms": 1e3 * min(host_s),
                    "pcie_only_ms": 1e3 * min(pcie_s), "pcie_gbs": pcie_bytes / min(pcie_s) / 1e9,
                    "frac_of_pcie": min(pcie_s) / min(host_s),
                    "note": "PyAscore.score_batch: host arrays in, host results out (chunked, upload pipelined "
                            "with planning, kernels and result copies); pcie_only = the same bytes copied up "
                            "and back with nothing else"}
        copy_gbs = achievable_hbm_gbs(torch, dev)
        default_size = args.psms is None and args.config != "cfg3" and args.max_charge is None
        traffic, traffic_src, valu_share, traffic_path = profiled_traffic(args.config, names[dom], default_size)
        line = {
            "metric": METRIC, "value": total * args.steps / elapsed, "unit": "PSMs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32/f64 scalar + i32 counts",
            "data": "synthetic (SURVEY.md 8(d) generator, job seed 1000, spectra per 16k-PSM block)",
            "config": {"workload": "%s: %d PSMs over %d GPU(s), work-balanced contiguous shards" % (args.config, total, world),
                       "psms_total": total, "psms_rank0": batch["n_psm"], "shard_sizes": [h - l for l, h in ranges], "peaks_per_spectrum": float(batch["peak_off"][-1]) / batch["n_psm"],
                       "signatures_total_per_gpu": plan.total_signatures, "mz_error": settings["mz_error"],
                       "fragment_types": settings["fragment_types"],
                       "max_fragment_charge": int(batch["max_charge"].max()),
                       "neutral_losses": settings["neutral_losses"], "parallelism": "psm-shard x%d + 1 gather" % world},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         # what actually bounds the kernel: share of its cycles in which the SIMDs' vector ALUs are
                         # occupied (SQ_ACTIVE_INST_VALU x 4 / SIMDs / busy cycles of the committed counter pass)
                         "valu_busy": valu_share,
                         # HBM bytes of the whole path per step (every kernel) next to the algorithmic bytes
                         "traffic_whole_path": traffic_path,
                         "traffic_over_algorithmic": (traffic_path / alg) if traffic_path else None,
                         "algorithmic_bytes_per_launch": alg,
                         "achievable_peak": copy_gbs, "frac_of_achievable": achieved / copy_gbs,
                         "whole_path_gbs": alg / (float(kern_ms.sum()) * 1e-3) / 1e9 if kern_ms.sum() > 0 else 0.0,
                         "kernel_ms": {n: float(m) for n, m in zip(names, kern_ms)}},
            "host_api": host,
            "workspace_bytes": plan.workspace_bytes,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, settings)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
