#!/bin/bash
# A/B of two in-tree builds on the same box: bash scripts/ab.sh "cfg2 cfg4" (base = pyascore_amd/libbase.so)
cd ${GRAFT_REPO_ROOT:-.}
for c in ${1:-cfg2}; do
  for rep in 1 2; do
    for lib in ${AB_LIBS:-libbase.so libpyascore_hip.so}; do
      PYA_LIB=$PWD/pyascore_amd/$lib python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 > /tmp/ab.json
      python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
print("$c $lib", "ms/step %.4f" % d["ms_per_step"], {k.replace("pya_","").replace("_kernel",""): round(v,4) for k,v in d["roofline"]["kernel_ms"].items()})
PY
    done
  done
done
