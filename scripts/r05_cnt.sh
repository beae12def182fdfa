#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ timeout 1200 python scripts/cnt_check.py 24 5 2>&1 | tail -60 | grep -v "checked (" 
  timeout 300 python scripts/big_check.py 64 21 2>&1 | tail -4
  bash scripts/r05_ab.sh r05i cfg5 "libpyascore_hip.so:PYA_DEBUG=32768 libpyascore_hip.so" | tail -4
} > gpurun_out/r05i_cnt.txt 2>&1
cat gpurun_out/r05i_cnt.txt
