#!/bin/bash
# r04: the fused kernel after its diet -- the whole GPU suite, a bench line, the truncation counters of the fused kernel
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --config cfg2 --no-cpu-baseline 2>gpurun_out/${1:-r04e}_bench_cfg2.err | tail -1 > gpurun_out/${1:-r04e}_bench_cfg2.json
python - <<PY
import json
d = json.load(open("gpurun_out/${1:-r04e}_bench_cfg2.json"))
print("cfg2 %.4g PSMs/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), d["roofline"]["kernel_ms"])
PY
bash scripts/trunc_pmc.sh cfg2 "40 41 42 43 44 45 46 47 0" pya_score_localize 2>&1 | tee gpurun_out/${1:-r04e}_trunc_fused.txt
