#!/bin/bash
# per-kernel durations of one config: bash scripts/kstat.sh <tag> <cfg> [extra bench args]
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD; TAG=$1; CFG=$2; shift; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; rm -rf /tmp/ks; mkdir -p /tmp/ks; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-host-api --no-other-configs "$@" > /tmp/ks/log.txt 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/${TAG}_kstat_$CFG.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("pya_"):
        print("%-46s calls %4s  avg %10.1f us  total %10.1f us" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
cat $ROOT/gpurun_out/${TAG}_kstat_$CFG.txt
