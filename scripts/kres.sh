#!/bin/bash
# registers / spills / scratch / LDS of every kernel of one source (CPU container, no GPU): bash scripts/kres.sh score_big [extra flags]
cd $(dirname $0)/../pyascore_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c $f.hip -o /tmp/kres_$f.bundle || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/kres_$f.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kres_$f.co || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kres_$f.co | python3 -c "
import sys, re
cur = {}
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count):\s*(\S+)', line)
    if not m: continue
    k, v = m.groups()
    if k == 'name' and not v.startswith('pya_') and not v.startswith('_Z'): continue
    cur[k] = v
    if len(cur) == 8:
        print('%-60s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  scratch %4s B  static LDS %s' % (cur['name'][:60], cur['vgpr_count'], cur['agpr_count'], cur['sgpr_count'], cur['vgpr_spill_count'], cur['sgpr_spill_count'], cur['private_segment_fixed_size'], cur['group_segment_fixed_size']))
        cur = {}
"
