#!/bin/bash
# Through gpurun: bash scripts/pmc.sh "<space separated counters>" [cfg] [lib.so ...]  -- one --pmc pass per library
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
CTRS=$1; CFG=${2:-cfg2}; shift; shift
LIBS=${@:-libpyascore_hip.so}
export TMPDIR=/tmp
cd /tmp
for lib in $LIBS; do
  rm -rf /tmp/pmc_$lib
  PYA_LIB=$ROOT/pyascore_amd/$lib timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmc_$lib -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $PYA_BENCH_EXTRA > /tmp/pmc_$lib.log 2>&1 || tail -5 /tmp/pmc_$lib.log
  echo "== $lib $CFG"
  python3 $ROOT/scripts/pmc_summary.py /tmp/pmc_$lib
done
