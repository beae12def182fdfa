#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
echo "== dist path of bench.py with a world of one"; PYA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config cfg2 --steps 5 --warmup 2 --blocks 2 --no-cpu-baseline --no-host-api 2>/dev/null | tail -1 | cut -c1-400
echo "== soak_general 0 240"; timeout 900 python scripts/soak_general.py 0 240 2>&1 | tail -4
echo "== soak_mixed 0 60"; timeout 600 python scripts/soak_mixed.py 0 60 2>&1 | tail -2
echo "== soak_adversarial 0 6000"; timeout 600 python scripts/soak_adversarial.py 0 6000 2>&1 | tail -2
echo "== fuzz 500:1100"; PYA_FUZZ_SEEDS=500:1100 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_soak.txt
