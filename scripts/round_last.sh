#!/bin/bash
# the suite twice (a flake shows), then the round's bench lines against the committed r04_z counters
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04_zz_gpu_tests_$rep.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_zz_gpu_tests_$rep.log | tail -2
done
bash scripts/profile.sh r04_zz "cfg1 cfg2 cfg3 cfg4 cfg5" "" > gpurun_out/r04_zz_profile.log 2>&1
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04_zz_bench_$c.json"))
    print("$c %.4g PSMs/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), {k.replace("pya_","").replace("_kernel",""): round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "frac %.4f" % d["roofline"]["frac"], d["roofline"].get("traffic_source"))
except Exception as e:
    print("$c", e)
PY
done
