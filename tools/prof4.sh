#!/bin/bash
# LDS / issue-stall PMC passes for the skin kernel
set -u
OUT=gpurun_out/$1; shift
rm -rf $OUT
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/bench.py $ARGS --no-cpu-baseline > $R/$OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$OUT/$name.log; return 1; }
}
run pmcA --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES || exit 1
run pmcB --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU || exit 1
python3 tools/prof_summary.py $OUT | grep -E "k_skin|=="
