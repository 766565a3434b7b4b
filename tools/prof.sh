#!/bin/bash
# Profiles `python bench.py ...` on the GPU box: kernel trace + PMC passes (one run each).
# usage: tools/prof.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=gpurun_out/$1; shift
rm -rf $OUT
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
run() { # name, rocprof args
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-extras > $R/$OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$OUT/$name.log; return 1; }
}
run trace --kernel-trace --stats || exit 1
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
run pmc2 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS || exit 1
run pmc3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
find $OUT -name "*.csv" | head -50
