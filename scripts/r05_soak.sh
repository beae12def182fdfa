#!/bin/bash
# round 5 soak: the count-node kernels and the new division on many seeds, beside the usual adversarial / fuzz soaks
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for seed in 11 12 13 14 15 16; do echo "== cnt_check seed $seed"; timeout 900 python scripts/cnt_check.py 16 $seed 2>&1 | tail -1; done
for seed in 41 42 43 44; do echo "== hash_check seed $seed"; timeout 900 python scripts/hash_check.py 120 $seed 2>&1 | tail -1; done
for seed in 51 52 53; do echo "== big_check seed $seed"; timeout 600 python scripts/big_check.py 48 $seed 2>&1 | tail -1; done
echo "== fuzz 500:1300"; PYA_FUZZ_SEEDS=500:1300 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
echo "== adversarial 0 1500"; timeout 1500 python scripts/soak_adversarial.py 0 1500 2>&1 | tail -3
echo "== handover stress x2"; PYA_STRESS_CALLS=120000 timeout 900 python -m pytest tests/test_gpu_handover_stress.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -3
} > gpurun_out/r05_soak.txt 2>&1
cat gpurun_out/r05_soak.txt
