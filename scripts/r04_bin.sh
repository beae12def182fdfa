#!/bin/bash
# r04: the composite-key binning -- the binning tests, a bench line, kernel stats + counters of cfg2
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tree or binning or equal_intensities or unsorted or public_api or peak_order or largest_spectra" 2>&1 | tail -4
bash scripts/profile.sh ${1:-r04c} "cfg2" "cfg2" 2>&1 | tail -40
python - <<PY
import json
d = json.load(open("gpurun_out/${1:-r04c}_bench_cfg2.json"))
print("cfg2 %.4g PSMs/s  %.3f ms/step" % (d["value"], d["ms_per_step"]), d["roofline"]["kernel_ms"])
PY
