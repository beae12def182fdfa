#!/bin/bash
# Scaling run on one node: bash scripts/scale.sh [config] [weak|strong] ["1 2 4 8"] [extra bench.py args]
# One process per GPU over RCCL (torch.distributed.run); one JSON line per N on stdout.
# Presets of BASELINE.json's multi-GPU configs (strong scaling, the job size fixed as N grows):
#   bash scripts/scale.sh cfg3 strong      1 M PSMs (L 8-40, 1-4 mods on <= 12 sites) over N GPUs
#   bash scripts/scale.sh cfg5 strong      50 k PSMs x 3003 site assignments over N GPUs ("1 vs 8 GPU")
# Every line carries multi_gpu.per_rank (kernel ms per family, gather wait, wall, shard size, work
# estimate per rank), so a curve that bends says where.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
CFG=${1:-cfg2}; MODE=${2:-weak}; NS=${3:-"1 2 4 8"}; shift; shift; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in $NS; do
  # bench.py launches its own ranks for N > 1 (one child process per GPU under torch.distributed.run)
  python3 bench.py --gpus $N --config $CFG --scaling $MODE --no-cpu-baseline --no-other-configs "$@" | tail -1
done
