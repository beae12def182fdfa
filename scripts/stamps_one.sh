#!/bin/bash
# Through gpurun: bash scripts/stamps_one.sh [cfg] -- phase cycles of one PSM alone (diagnostic build, then the normal one back)
cd ${GRAFT_REPO_ROOT:-.}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_BUILD_STAMPS=1 python -m pyascore_amd.build --force > /tmp/stamps_build.log 2>&1 || tail -5 /tmp/stamps_build.log
PYA_STAMPS=1 python scripts/stamps_one.py ${1:-cfg2} 200 2>&1 | grep -E "pya stamps|runs" | sort -k4 -n
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
