#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash scripts/trunc_pmc.sh cfg5 "13 14 15 16 9 10 11 12 0" pya_score_big_kernel > gpurun_out/r05k_truncpmc_cfg5.txt 2>&1
bash scripts/trunc.sh cfg5 "13 14 15 16 9 10 11 12" >> gpurun_out/r05k_truncpmc_cfg5.txt 2>&1
cat gpurun_out/r05k_truncpmc_cfg5.txt
