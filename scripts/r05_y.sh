#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "batch_matches_checker or fuzz or golden or window_edges or baseline or tied or equal" 2>&1 | tail -2
  bash scripts/r05_ab.sh r05y cfg3 "libpyascore_hip.so" | tail -2
  bash scripts/r05_ab.sh r05y cfg2 "libpyascore_hip.so" | tail -2
} > gpurun_out/r05y.txt 2>&1
cat gpurun_out/r05y.txt
