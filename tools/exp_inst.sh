#!/bin/bash
# skin stage time per 10,000 instances for crowds of different sizes (C3 rig): tools/exp_inst.sh 2500 5000 10000 ...
for n in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c3 --instances $n > gpurun_out/b_inst_$n.log 2> gpurun_out/b_inst_$n.err || { echo "$n failed"; tail -3 gpurun_out/b_inst_$n.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_inst_$n.log").read().strip().splitlines()[-1]); s=d["stage_ms"]["skin_morph"]
print("$n instances: skin", round(s,4), "ms =", round(s*10000/$n,4), "per 10k; frac", round(d["roofline"]["frac"],4), "placement", d["output_placement"]["strategy"], round(d["output_placement"]["probe_GBps"]))
PY
done
