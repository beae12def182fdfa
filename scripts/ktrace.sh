#!/bin/bash
# Through gpurun: bash scripts/ktrace.sh cfg5 -- one kernel-trace pass, per-dispatch resources and durations of the second run
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $ROOT/scripts/trunc_run.py ${1:-cfg2} ${2:-0} > /tmp/kt.log 2>&1 || tail -3 /tmp/kt.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "pya_" in r["Kernel_Name"]]
    half = rows[len(rows) // 2:]                 # the second of the two runs
    for r in half:
        print("%-44s lds %6s scratch %4s vgpr %3s grid %9s  %.3f ms" % (
            r["Kernel_Name"][:44], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Grid_Size_X"],
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
