#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_gpu_count_nodes.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "batch_matches_checker or charges or general or mimic or golden" 2>&1 | tail -2
  python scripts/hash_check.py 150 31 2>&1 | tail -1
  bash scripts/r05_ab.sh r05w cfg4 "libpyascore_hip.so:PYA_NO_CNT=1 libpyascore_hip.so" | tail -4
} > gpurun_out/r05w.txt 2>&1
cat gpurun_out/r05w.txt
