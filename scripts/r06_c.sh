#!/bin/bash
# r06: the selection route for dense spectra (tests, then the dense legs), score_big after the lazy residue mask
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-r06_c}
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_dense.py tests/test_gpu_count_nodes.py -m gpu -x -q -p no:cacheprovider --durations=5 > gpurun_out/${TAG}_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests.txt
tail -12 gpurun_out/${TAG}_tests.txt
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_realistic.py -m gpu -x -q -p no:cacheprovider -k "binning or equal_intens or unsorted or full_size or largest or realistic_clusters_match_the_reference and fused" > gpurun_out/${TAG}_tests2.txt 2>&1
echo "rc=$?" >> gpurun_out/${TAG}_tests2.txt
tail -5 gpurun_out/${TAG}_tests2.txt
bash scripts/abx.sh $TAG cfg5 "libpyascore_hip.so libpyascore_hip.so:PYA_DEBUG=268435456"
for leg in dense1500 dense4000; do
  for sw in "" "--debug PYA_BIN_SELECT_MIN=1000000"; do
    python bench.py --leg $leg --steps 20 --warmup 3 --other-blocks 2 $sw 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$leg $sw', 'ms/step %.4f' % d['ms_per_step'], 'bin ns/peak %.4f' % d.get('bin_ns_per_peak', -1), {k.replace('pya_','').replace('_kernel',''): round(v,4) for k,v in d['kernel_ms'].items()})" 2>&1 | tail -1
  done
done | tee gpurun_out/${TAG}_dense.txt
