#!/bin/bash
# Through gpurun: bash scripts/trunc_pmc.sh cfg2 "38 39 48 ..." [kernel-name-substring]
# Diagnostic build; one counter pass per truncation point: vector / scalar / LDS instructions per wavefront
# up to each stamp of the kernel (differences = the phases).
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
CFG=${1:-cfg2}; PTS=${2:-"38 39 48 49 50 40 41 42 43 44 45 46 47 0"}; KN=${3:-pya_score_localize}
cp pyascore_amd/libpyascore_hip.so /tmp/keep.so
PYA_BUILD_STAMPS=1 python -m pyascore_amd.build --force > /tmp/stamps_build.log 2>&1 || tail -5 /tmp/stamps_build.log
export TMPDIR=/tmp
cd /tmp
for k in $PTS; do
  rm -rf /tmp/tp_$k
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/tp_$k -- python3 $ROOT/scripts/trunc_run.py $CFG $((k << 16)) > /tmp/tp_$k.log 2>&1 || tail -3 /tmp/tp_$k.log
  python3 - "$k" "$KN" <<'PY'
import csv, glob, sys, collections
k, kn = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
n = 0
for f in glob.glob("/tmp/tp_%s/**/*counter_collection.csv" % k, recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
w = acc.get("SQ_WAVES", 0) or 1
print("stop %3s: per wave VALU %7.1f SALU %7.1f LDS %6.1f" % (k, acc["SQ_INSTS_VALU"] / w, acc["SQ_INSTS_SALU"] / w, acc["SQ_INSTS_LDS"] / w))
PY
done
cd $ROOT
cp /tmp/keep.so pyascore_amd/libpyascore_hip.so
python -m pyascore_amd.build --force > /dev/null 2>&1
