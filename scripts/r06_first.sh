#!/bin/bash
# r06, first GPU call: the new tests (realistic clusters, pack kernel), then the default bench line with the dense legs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r06_a}
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_realistic.py "tests/test_gpu_parity.py::test_gather_records_are_packed_by_the_library" "tests/test_gpu_parity.py::test_loaded_library_is_built_from_this_tree" -m gpu -x -q -p no:cacheprovider --durations=8 > gpurun_out/${TAG}_new_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_new_tests.txt
tail -25 gpurun_out/${TAG}_new_tests.txt
( time timeout 1200 python bench.py ) 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_cfg2.json
tail -5 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_cfg2.json"))
print("cfg2 %.4g PSMs/s %.3f ms plan_ms %.1f whole_step_frac %.3f" % (d["value"], d["ms_per_step"], d["plan_ms"], d["roofline"]["whole_step_frac"]), d["roofline"]["kernel_ms"])
for k, v in d.get("other_configs", {}).items():
    if "error" in v: print(k, v); continue
    print(k, "%.4g PSMs/s %.3f ms plan %.1f ms peaks %.0f bin ns/peak %.3f" % (v["value"], v["ms_per_step"], v["plan_ms"], v["peaks_per_spectrum"], v["bin_ns_per_peak"]), {a.replace("pya_","").replace("_kernel",""): round(b,3) for a,b in v["kernel_ms"].items()}, "wall %.1f" % v.get("leg_wall_s", 0))
PY
