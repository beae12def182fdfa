#!/bin/bash
# Through gpurun: the GPU suite on a build whose DPP wave helpers trap when entered with a partial EXEC mask
# (device_common.hip.h: PYA_CHECK_EXEC; r05 advisor).  bash scripts/r06_check_exec.sh [tag]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r06_check_exec}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_DEFS=-DPYA_CHECK_EXEC python -m pyascore_amd.build --force > /tmp/ce_build.log 2>&1 || tail -5 /tmp/ce_build.log
{
echo "build: PYA_DEFS=-DPYA_CHECK_EXEC (every wave_sum / wave_max / wave_min / scan helper traps on a partial EXEC mask)"
python -c 'from pyascore_amd import _lib; print(_lib.load().pya_version().decode())'
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_loaded_library_is_built_from_this_tree 2>&1 | grep -E "passed|failed|error|Error|trap|Fatal" | tail -6
} > gpurun_out/${TAG}.txt 2>&1
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
cat gpurun_out/${TAG}.txt
