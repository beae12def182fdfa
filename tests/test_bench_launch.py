"""`python bench.py --gpus N` as the driver writes it, for N > 1, on a box without a GPU: bench.py must start its own
ranks (children under torch.distributed.run, before anything touches a GPU), rendezvous on 127.0.0.1, run the step
loop's gathers and print ONE JSON line.  PYA_BENCH_BACKEND=gloo makes the ranks dry-run the plumbing (nothing is
scored: `value` is null) -- the launch, the partition, shard.StepPipeline and the line's shape are what is under test.
The scoring itself under N > 1 is tests/test_sharding_gloo.py (CPU) and the -m gpu RCCL test."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, n):
    env = dict(os.environ, PYA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [r for r in p.stdout.splitlines() if r.strip()]
    assert len(rows) == 1, p.stdout                     # one line on stdout and nothing else
    return json.loads(rows[0])


def test_two_ranks_launched_by_bench_itself():
    line = _run(["--config", "cfg3", "--psms", "300"], 2)
    assert line["n_gpus"] == 2 and line["value"] is None and "DRY RUN" in line["data"]
    m = line["multi_gpu"]
    assert m["rccl_ranks"] == 2 and m["backend"] == "gloo"
    assert m["gathered_in_input_order"] is True and m["steps_gathered"] == 4
    sizes = line["config"]["shard_sizes"]
    assert len(sizes) == 2 and sum(sizes) == 600 and min(sizes) > 0


def test_strong_scaling_over_three_ranks():
    line = _run(["--config", "cfg5", "--scaling", "strong", "--total", "90"], 3)
    assert line["multi_gpu"]["rccl_ranks"] == 3 and sum(line["config"]["shard_sizes"]) == 90
    assert line["multi_gpu"]["gathered_in_input_order"] is True


def test_a_rank_count_that_disagrees_with_the_environment_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
