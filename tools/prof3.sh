#!/bin/bash
# L2 / HBM traffic passes for the skin kernel
set -u
OUT=gpurun_out/$1; shift
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py $ARGS --no-cpu-baseline > $R/$OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$OUT/$name.log; return 1; }
}
run pmc3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc8 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum || exit 1
run pmc9 --pmc TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum || exit 1
