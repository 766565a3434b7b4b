#!/bin/bash
# bench the default library on several workloads: tools/exp_wl.sh c3 c5 ...  (stage times and roofline fraction per workload)
for wl in "$@"; do
  timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --workload $wl > gpurun_out/b_wl_$wl.log 2> gpurun_out/b_wl_$wl.err || { echo "$wl failed rc=$?"; tail -5 gpurun_out/b_wl_$wl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_wl_$wl.log").read().strip().splitlines()[-1]);print("$wl",{k: round(v,4) for k,v in d["stage_ms"].items()},"ms_per_step",round(d["ms_per_step"],4),"frac",round(d["roofline"]["frac"],4))
PY
done
