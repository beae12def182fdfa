#!/bin/bash
# Scaling run on one node: bash scripts/scale.sh [config] [weak|strong] ["1 2 4 8"] [extra bench.py args]
# One process per GPU over RCCL (torch.distributed.run); one JSON line per N on stdout.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
CFG=${1:-cfg2}; MODE=${2:-weak}; NS=${3:-"1 2 4 8"}; shift; shift; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in $NS; do
  if [ "$N" = "1" ]; then
    python3 bench.py --gpus 1 --config $CFG --scaling $MODE --no-cpu-baseline "$@" | tail -1
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --config $CFG --scaling $MODE --no-cpu-baseline "$@" | tail -1
  fi
done
