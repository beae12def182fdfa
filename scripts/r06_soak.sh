#!/bin/bash
# round 6 soak: what the round changed (score_big's candidate records, the selection route of the binning, byte-count nodes and
# the turn-taking work areas of score_cntg, the descriptor prologues, the lean localize layout) on many seeds, beside the usual
# adversarial / fuzz / hash soaks:  bash scripts/r06_soak.sh [tag]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r06_soak}
{
echo "library: $(python -c 'from pyascore_amd import _lib; print(_lib.load().pya_version().decode())')"
for seed in 21 22 23 24 25 26; do echo "== big_check seed $seed (cfg5 x 48, cfg3 x 960)"; timeout 600 python scripts/big_check.py 48 $seed 2>&1 | tail -1; done
echo "== big_check without candidate records (PYA_DEBUG=0x10000000), seed 27"; PYA_DEBUG=268435456 timeout 600 python scripts/big_check.py 48 27 2>&1 | tail -1
for seed in 31 32 33 34; do echo "== cnt_check seed $seed"; timeout 900 python scripts/cnt_check.py 16 $seed 2>&1 | tail -1; done
echo "== dense_check seeds 0..60"; timeout 1500 python scripts/dense_check.py 0 60 2>&1 | tail -3
for seed in 71 72 73 74; do echo "== hash_check seed $seed"; timeout 900 python scripts/hash_check.py 120 $seed 2>&1 | tail -1; done
echo "== fuzz 3000:3800"; PYA_FUZZ_SEEDS=3000:3800 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
echo "== adversarial 0 2000"; timeout 1500 python scripts/soak_adversarial.py 0 2000 2>&1 | tail -3
echo "== handover stress x2"; PYA_STRESS_CALLS=120000 timeout 900 python -m pytest tests/test_gpu_handover_stress.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | tail -3
} > gpurun_out/${TAG}.txt 2>&1
cat gpurun_out/${TAG}.txt
