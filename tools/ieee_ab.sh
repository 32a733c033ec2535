#!/bin/bash
# developer: the IEEE walk limit (tools/ab/libie*.so) through bench.py's ieee leg
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05
for l in ${@:-base ie6 ie10 ie12}; do
  WSFLUID_LIBRARY=$PWD/tools/ab/lib$l.so python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-north-star --no-cpu-baseline --no-readback --reps 3 > gpurun_out/r05/ieee_$l.json 2> gpurun_out/r05/ieee_$l.err
  python3 - "$l" <<'PY'
import json, sys
l = sys.argv[1]
d = json.loads(open("gpurun_out/r05/ieee_%s.json" % l).read().strip().splitlines()[-1]); i = d["ieee"]
print(l, round(d["ms_per_step"], 4), round(d["settled"]["ms_per_step"], 4), "ieee", round(i["ms_per_step"], 4), round(i["settled"]["ms_per_step"], 4), i["settled"]["kernel_ms"])
PY
done
