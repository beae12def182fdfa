#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ timeout 600 python scripts/big_check.py 64 21 2>&1 | tail -3
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "batch_matches_checker or baseline or golden or window_edges or fuzz" 2>&1 | tail -3
  bash scripts/r05_ab.sh r05n cfg3 "libpyascore_hip.so:PYA_NO_CNT=1 libpyascore_hip.so" | tail -4
} > gpurun_out/r05n_cnt3.txt 2>&1
cat gpurun_out/r05n_cnt3.txt
