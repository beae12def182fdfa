#!/bin/bash
# Through gpurun: bash scripts/stamps_realistic.sh -- the rare-branch counters on the realistic-cluster data (and on cfg4 / cfg5 beside them)
cd ${GRAFT_REPO_ROOT:-.}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_BUILD_STAMPS=1 python -m pyascore_amd.build --force > /tmp/stamps_build.log 2>&1 || tail -5 /tmp/stamps_build.log
for f in general plain; do
  echo "== realistic $f (2048 PSMs, one run)"
  pat="phase (4[0-4]|5[6-9]|62):"; [ $f = plain ] && pat="phase (5[0-9]):"      # (the fused kernel's time stamps use 38-50)
  PYA_STAMPS=1 python scripts/stamps_realistic.py $f 2>&1 | grep "pya stamps" | grep -E "$pat"
done
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
