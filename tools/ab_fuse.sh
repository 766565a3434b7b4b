#!/bin/bash
# A/B of the skin stage: k_skin_fix next to k_skin2 on a side stream (enqueue order, priority, grid) vs one after the other
set -e
out=gpurun_out/r2_fuse.log
: > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>>$out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], d['ms_per_step'], d['stage_ms'])" >> $out; }
run SKINRT_FIX_STREAM=0
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=1
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=1 SKINRT_FIX_PRIO=1
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=1 SKINRT_FIX_PRIO=-1
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=0 SKINRT_FIX_PRIO=1
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=0 SKINRT_FIX_PRIO=-1
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=1 SKINRT_FIX_BLOCKS=8192
run SKINRT_FIX_STREAM=1 SKINRT_FIX_ORDER=1 SKINRT_FIX_BLOCKS=4096 SKINRT_FIX_PRIO=-1
run SKINRT_FIX_STREAM=0
cat $out
