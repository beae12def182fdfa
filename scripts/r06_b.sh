#!/bin/bash
# r06: score_big's candidate mode -- tests that reach it, then cfg5 with and without it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r06_b}
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_count_nodes.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "count or big or full_size or golden or cfg5 or ties or many_tied or sort" --durations=5 > gpurun_out/${TAG}_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.txt
tail -12 gpurun_out/${TAG}_tests.txt
bash scripts/abx.sh $TAG cfg5 "libpyascore_hip.so libpyascore_hip.so:PYA_DEBUG=268435456"
bash scripts/kstat.sh $TAG cfg5
