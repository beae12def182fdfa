#!/bin/bash
# Through gpurun: bash scripts/profile.sh <tag> ["cfg list for bench lines"] ["cfg list to profile"]
# Writes gpurun_out/<tag>_bench_<cfg>.json and, per profiled config,
# gpurun_out/<tag>_rocprof_<cfg>/{kernel_stats.csv,pmc_summary.csv}.
# Counter passes are separate runs with --pmc only (never combined with trace domains).
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r02_x}
CFGS=${2-"cfg1 cfg2 cfg3 cfg4 cfg5"}
PCFGS=${3-"cfg2"}
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for c in $CFGS; do
  extra="--no-cpu-baseline"; [ "$c" = "cfg2" ] && extra=""
  timeout 900 python3 bench.py --config $c $extra 2>$OUT/${TAG}_bench_$c.err | tail -1 > $OUT/${TAG}_bench_$c.json
  [ -s $OUT/${TAG}_bench_$c.json ] && rm -f $OUT/${TAG}_bench_$c.err
done
for PCFG in $PCFGS; do
  P=$OUT/${TAG}_rocprof_$PCFG
  rm -rf $P /tmp/prof && mkdir -p $P /tmp/prof
  cd /tmp
  STEPS=10; WARM=2; PSTEPS=3; PWARM=1
  SRC=$(PYTHONPATH=$ROOT python3 -c "from pyascore_amd import _lib; print(_lib.load().pya_version().decode().split('src=')[-1])")
  echo "{\"stats_steps\": $((STEPS+WARM)), \"pmc_steps\": $((PSTEPS+PWARM)), \"src\": \"$SRC\"}" > $P/profile_steps.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -- python3 $ROOT/bench.py --config $PCFG --steps $STEPS --warmup $WARM --blocks 1 --no-cpu-baseline --no-host-api --no-other-configs > /tmp/prof/stats.log 2>&1
  f=$(find /tmp/prof/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/kernel_stats.csv
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/prof/pmc$i -- python3 $ROOT/bench.py --config $PCFG --steps $PSTEPS --warmup $PWARM --blocks 1 --no-cpu-baseline --no-host-api --no-other-configs > /tmp/prof/pmc$i.log 2>&1 || tail -3 /tmp/prof/pmc$i.log
  done
  cd $ROOT
  python3 scripts/pmc_summary.py /tmp/prof $((PSTEPS+PWARM)) > $P/pmc_summary.csv
  echo "== $PCFG"; cat $P/kernel_stats.csv | head -12; cat $P/pmc_summary.csv
done
