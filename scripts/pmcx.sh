#!/bin/bash
# counters of one config's kernels, per dispatch: bash scripts/pmcx.sh <tag> <cfg> [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD; TAG=$1; CFG=$2; shift; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
rm -rf /tmp/pm; mkdir -p /tmp/pm
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pm/p$i -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-host-api --no-other-configs "$@" > /tmp/pm/p$i.log 2>&1 || tail -3 /tmp/pm/p$i.log
done
cd $ROOT
python3 scripts/pmc_summary.py /tmp/pm 4 > gpurun_out/${TAG}_pmc_$CFG.csv
python3 - gpurun_out/${TAG}_pmc_$CFG.csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["kernel"]][r["counter"]] = float(r["per_step"])
for k, c in d.items():
    w = c.get("SQ_WAVES", 0) or 1
    cyc = c.get("SQ_BUSY_CYCLES", 0) / 32.0 or 1
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print("%-42s waves %8d  VALU/wave %8.1f SALU/wave %8.1f LDS/wave %7.1f  lanes/VALU %4.1f  valu_issue(x4cyc) %.2f salu %.2f lds_active %.2f bankconf/lds %.2f wait_any/wavecyc %.2f"
          % (k[:42], w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w,
             c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1), c.get("SQ_INSTS_VALU", 0) * 4.0 / (1024.0 * cyc),
             c.get("SQ_INSTS_SALU", 0) / (256.0 * cyc), c.get("SQ_LDS_IDX_ACTIVE", 0) / (256.0 * cyc),
             c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
