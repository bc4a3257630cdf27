#!/bin/bash
# Counter passes over scratch/ibl_one.py (one launch of each IBL kernel): traffic, L2, lanes, texture addresser.
# usage: bash scratch/pmc_ibl3.sh <out-subdir-of-gpurun_out>
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
run c FETCH_SIZE TCC_HIT_sum && \
run d WRITE_SIZE TCC_MISS_sum TCC_REQ_sum && \
run e SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS && \
run f TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
echo passes done
