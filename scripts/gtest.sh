#!/bin/bash
# run selected GPU tests with the full log kept:  bash scripts/gtest.sh <tag> <pytest args...>
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=$1; shift
timeout 1500 python -X faulthandler -m pytest "$@" -p no:cacheprovider > gpurun_out/${TAG}.txt 2>&1
echo "rc=$?" >> gpurun_out/${TAG}.txt
grep -n "Fatal\|fault\|Error\|passed\|failed\|rc=" gpurun_out/${TAG}.txt | head -20
grep -n "File \"/root/repo\|File \".*tests/" gpurun_out/${TAG}.txt | head -12
