#!/bin/bash
# the round's record: whole GPU suite, then scripts/profile.sh (bench lines for every config, kernel stats + counters)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r04_z}
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
bash scripts/profile.sh $TAG "cfg1 cfg2 cfg3 cfg4 cfg5" "cfg2 cfg3 cfg4 cfg5" > gpurun_out/${TAG}_profile.log 2>&1
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_$c.json"))
    print("$c %.4g PSMs/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), d["blocks"]["ms_per_step"], {k.replace("pya_","").replace("_kernel",""): round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "frac %.4f" % d["roofline"]["frac"], "host_api %.3g" % (d["host_api"] or {}).get("value", 0))
except Exception as e:
    print("$c", e)
PY
done
