#!/bin/bash
# a second, longer soak on the round's final kernels: other seeds than scripts/r04_soak.sh
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
echo "== soak_general 240 600"; timeout 1500 python scripts/soak_general.py 240 600 2>&1 | tail -2
echo "== soak_mixed 60 200"; timeout 1500 python scripts/soak_mixed.py 60 200 2>&1 | tail -2
echo "== soak_adversarial 6000 20000"; timeout 1200 python scripts/soak_adversarial.py 6000 20000 2>&1 | tail -1
echo "== fuzz 1100:2300"; PYA_FUZZ_SEEDS=1100:2300 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
echo "== big_check 2000 11"; timeout 900 python scripts/big_check.py 2000 11 2>&1 | tail -2
echo "== hash_check 3000 5"; timeout 900 python scripts/hash_check.py 3000 5 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_soak2.txt
