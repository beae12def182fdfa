#!/bin/bash
# r06 generic: bash scripts/r06_ab.sh <tag> "<pytest args>" <cfg> "<libs for abx>" ["<-k expression>"]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=$1; PT=$2; CFG=$3; LIBS=$4; KEXPR=$5
if [ -n "$PT" ]; then
  timeout 2400 python -X faulthandler -m pytest $PT ${KEXPR:+-k "$KEXPR"} -m gpu -x -q -p no:cacheprovider --durations=5 > gpurun_out/${TAG}_tests.txt 2>&1
  echo "rc=$?" >> gpurun_out/${TAG}_tests.txt
  tail -12 gpurun_out/${TAG}_tests.txt
fi
[ -n "$CFG" ] && bash scripts/abx.sh $TAG $CFG "$LIBS"
