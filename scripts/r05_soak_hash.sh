#!/bin/bash
# round 5, after the closed forms of the hash route: its parity probe on many seeds + the adversarial soak (general-only route included)
#   bash scripts/r05_soak_hash.sh [first hash_check seed] [seeds] [adversarial seeds]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S0=${1:-61}; NS=${2:-6}; ADV=${3:-2500}
{
for seed in $(seq $S0 $((S0 + NS - 1))); do echo "== hash_check seed $seed"; timeout 900 python scripts/hash_check.py 150 $seed 2>&1 | tail -1; done
echo "== adversarial 0 $ADV"; timeout 1500 python scripts/soak_adversarial.py 0 $ADV 2>&1 | tail -3
echo "== fuzz 2000:2400"; PYA_FUZZ_SEEDS=2000:2400 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
} > gpurun_out/r05_soak_hash.txt 2>&1
cat gpurun_out/r05_soak_hash.txt
