#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/scratch/ibl_one.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY && \
run c FETCH_SIZE TCC_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum && \
run d TCC_MISS_sum TCC_REQ_sum
echo passes done
