#!/bin/bash
# Through gpurun: bash scripts/trunc.sh cfg2 ["48 49 50 40 ..."] -- diagnostic build, cumulative time profile by truncation
cd ${GRAFT_REPO_ROOT:-.}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_BUILD_STAMPS=1 python -m pyascore_amd.build --force > /tmp/stamps_build.log 2>&1 || tail -5 /tmp/stamps_build.log
python scripts/trunc_probe.py ${1:-cfg2} "${2:-48 49 50 40 41 42 43 44 45 46 47}" 2>&1 | grep "stop at"
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
