#!/bin/bash
# cfg4: cumulative time profile by truncation, then instruction counts per truncation point for the two big kernels
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash scripts/trunc.sh cfg4 "1 2 3 4 5 6 7 8 20 21 22 23 24 25 26 27 28 30 31 32 33 34 35 36" > gpurun_out/r05d_trunc_cfg4.txt 2>&1
bash scripts/trunc_pmc.sh cfg4 "20 24 25 27 28 30 31 32 34 35 36 0" pya_localize_hash > gpurun_out/r05d_truncpmc_cfg4_loc.txt 2>&1
bash scripts/trunc_pmc.sh cfg4 "1 2 3 4 5 6 7 8 0" pya_score_nodes > gpurun_out/r05d_truncpmc_cfg4_score.txt 2>&1
cat gpurun_out/r05d_trunc_cfg4.txt gpurun_out/r05d_truncpmc_cfg4_loc.txt gpurun_out/r05d_truncpmc_cfg4_score.txt
