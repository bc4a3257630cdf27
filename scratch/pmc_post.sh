#!/bin/bash
# Counter passes over the one-kernel bloom chain.  usage: bash scratch/pmc_post.sh <out-subdir-of-gpurun_out>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/scratch/post_one.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY && \
run sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT && \
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_BRANCH && \
run mem FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE && \
run mem2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
python3 $R/scratch/pmc_summary.py $OUT bloom_post_fused
