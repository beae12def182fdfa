#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash scripts/trunc_pmc.sh cfg4 "30 60 61 31 32 0" pya_localize_hash > gpurun_out/r05h_truncpmc_cfg4_loc.txt 2>&1
PYA_DEBUG=2 bash scripts/trunc_pmc.sh cfg4 "32 0" pya_localize_hash >> gpurun_out/r05h_truncpmc_cfg4_loc.txt 2>&1
cat gpurun_out/r05h_truncpmc_cfg4_loc.txt
