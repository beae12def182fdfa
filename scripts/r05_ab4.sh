#!/bin/bash
# cfg4 A/B: the split general localize route against the one-kernel route, and the sites kernel at 4 / 5 / 6 wavefronts per SIMD
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
python scripts/hash_check.py 150 31 2>&1 | tail -4
one() { PYA_LIB=$PWD/pyascore_amd/$1 python bench.py --config cfg4 --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-host-api --no-other-configs $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2', 'ms/step %.3f' % d['ms_per_step'], {k.replace('pya_','').replace('_kernel',''): round(v,3) for k,v in d['roofline']['kernel_ms'].items()})"; }
for rep in 1 2; do
  one libpyascore_hip.so "--debug PYA_NO_LOC_SPLIT=1"
  one libpyascore_hip.so ""
  one lib_w4.so ""

done
} > gpurun_out/r05e_ab4.txt 2>&1
cat gpurun_out/r05e_ab4.txt
