#!/bin/bash
# Usage (through gpurun): bash scripts/gpu_check.sh "cfg2 cfg5" [pytest-args]
# Runs the GPU parity tests, then bench.py for the named configs; logs under gpurun_out/.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
CFGS=${1:-cfg2}
shift
timeout 1200 python -m pytest tests -m gpu -x -q "$@" 2>&1 > gpurun_out/gpu_tests.log; grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -5
for c in $CFGS; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 > gpurun_out/bench_$c.json
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_$c.json"))
print("$c", "%.3g PSMs/s" % d["value"], "ms/step %.3f" % d["ms_per_step"], {k.replace("pya_","").replace("_kernel",""): round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "host_api %.3g" % d["host_api"]["value"])
PY
done
