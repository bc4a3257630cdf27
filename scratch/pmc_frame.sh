#!/bin/bash
# Counter passes over the frame chain of bench.py (shade, bloom H, fused bloom V + post-process).
# usage: bash scratch/pmc_frame.sh <out-subdir-of-gpurun_out>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-ibl --no-coverage --no-ranges-compare --cpu-rows 0 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run fetch FETCH_SIZE TCC_HIT_sum && \
run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum && \
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run ta SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
echo passes done
