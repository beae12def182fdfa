#!/bin/bash
# Through gpurun: bash scripts/stamps.sh "cfg2 cfg4" [max_charge] -- diagnostic build with in-kernel phase stamps (shares, not lengths)
cd ${GRAFT_REPO_ROOT:-.}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_BUILD_STAMPS=1 python -m pyascore_amd.build --force > /tmp/stamps_build.log 2>&1 || tail -5 /tmp/stamps_build.log
for c in ${1:-cfg2}; do
  echo "== $c $2"
  PYA_STAMPS=1 python scripts/stamps_probe.py $c $2 2>&1 | grep "pya stamps" | sort -k4 -n
done
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
