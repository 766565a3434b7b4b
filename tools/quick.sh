#!/bin/bash
# quick GPU check: parity tests then a short bench; prints stage ms and roofline fraction
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread "$@" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; tail -3 gpurun_out/pytest_gpu.log
# anything but pass / ordinary test failure (fault, abort, time limit): stop here, no more GPU work
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for wl in ${WL:-c3}; do
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $wl > gpurun_out/b_$wl.log 2>&1; echo bench $wl rc=$?
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_$wl.log").read().strip().splitlines()[-1]);print("$wl",d["stage_ms"],"frac",round(d["roofline"]["frac"],4),"value",d["value"])
except Exception as e:
    print(open("gpurun_out/b_$wl.log").read()[-2000:])
PY
done
