#!/bin/bash
# round 5, first GPU call: the whole GPU suite, the new tests verbosely, the bench line with other_configs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r05a_suite.txt
timeout 600 python -m pytest tests/test_gpu_handover_stress.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r05a_stress.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r05a_bench.err | tail -1 > gpurun_out/r05a_bench.json
cat gpurun_out/r05a_suite.txt gpurun_out/r05a_stress.txt; cut -c1-600 gpurun_out/r05a_bench.json; tail -3 gpurun_out/r05a_bench.err
