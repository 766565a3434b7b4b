#!/bin/bash
# extra PMC passes for the store path
set -u
OUT=gpurun_out/$1; shift
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py $ARGS --no-cpu-baseline > $R/$OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$OUT/$name.log; return 1; }
}
run pmc5 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES || exit 1
run pmc6 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum || exit 1
run pmc7 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 1
