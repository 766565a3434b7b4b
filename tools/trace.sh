#!/bin/bash
# kernel trace only (per-kernel average duration) of `python bench.py ...`: tools/trace.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/$OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $R/$OUT/trace.log; exit 1; }
python3 tools/prof_summary.py $OUT
