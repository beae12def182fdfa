#!/bin/bash
# A/B of in-tree library builds on one box: bash scripts/abx.sh <tag> <cfg> "<lib[:debug switches]> ..."   (2 rounds)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=$1; CFG=$2; LIBS=$3
one() { PYA_LIB=$PWD/pyascore_amd/$1 python bench.py --config $CFG --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-host-api --no-other-configs $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG $1 $2', 'ms/step %.4f' % d['ms_per_step'], {k.replace('pya_','').replace('_kernel',''): round(v,4) for k,v in d['roofline']['kernel_ms'].items()})"; }
{
for rep in 1 2; do
  for spec in $LIBS; do
    lib=${spec%%:*}; dbg=""; [ "$spec" != "$lib" ] && dbg="--debug ${spec#*:}"
    one $lib "$dbg"
  done
done
} > gpurun_out/${TAG}_ab_$CFG.txt 2>&1
cat gpurun_out/${TAG}_ab_$CFG.txt
