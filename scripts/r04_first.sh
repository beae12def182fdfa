#!/bin/bash
# r04 first GPU call: the suite on the HEAD build, bench lines, truncation counter profiles of cfg2's two kernels
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 > gpurun_out/r04a_gpu_tests.log; tail -3 gpurun_out/r04a_gpu_tests.log
for c in cfg2 cfg4 cfg5; do
  timeout 600 python bench.py --config $c --no-cpu-baseline 2>gpurun_out/r04a_bench_$c.err | tail -1 > gpurun_out/r04a_bench_$c.json
done
bash scripts/trunc_pmc.sh cfg2 "38 39 48 49 50 40 41 42 43 44 45 46 47 0" pya_score_localize > gpurun_out/r04a_trunc_fused.txt 2>&1
bash scripts/trunc_pmc.sh cfg2 "1 2 3 4 0" pya_bin_spectra > gpurun_out/r04a_trunc_bin.txt 2>&1
cat gpurun_out/r04a_trunc_fused.txt gpurun_out/r04a_trunc_bin.txt
