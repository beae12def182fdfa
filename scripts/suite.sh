#!/bin/bash
# the whole GPU suite (summary line only) and bench lines for the named configs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r04x}; CFGS=${2:-"cfg2"}
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/${TAG}_gpu_tests.log | head -20
for c in $CFGS; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-other-configs 2>gpurun_out/${TAG}_bench_$c.err | tail -1 > gpurun_out/${TAG}_bench_$c.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_$c.json"))
print("$c %.4g PSMs/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), {k.replace("pya_","").replace("_kernel",""): round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "host_api %.3g" % d["host_api"]["value"])
PY
done
