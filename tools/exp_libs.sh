#!/bin/bash
# bench the C3 workload with several builds of the library (SKINRT_LIB selects the .so)
# usage: tools/exp_libs.sh name1 name2 ...   (renderer_amd/lib/libskinrt_<name>.so; "default" = libskinrt.so)
for n in "$@"; do
  if [ "$n" = default ]; then lib=renderer_amd/lib/libskinrt.so; else lib=renderer_amd/lib/libskinrt_$n.so; fi
  SKINRT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload ${WL:-c3} > gpurun_out/b_$n.log 2> gpurun_out/b_$n.err || { echo "$n failed rc=$?"; tail -5 gpurun_out/b_$n.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1]);print("$n",d["stage_ms"],"frac",round(d["roofline"]["frac"],4))
PY
  grep -h "skinrt plan\|stamp" gpurun_out/b_$n.err | tail -30
done
