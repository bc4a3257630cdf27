#!/bin/bash
# Kernel-trace statistics of a short benchmark run with the library in VKR_HIP_LIB.  usage: bash scratch/profile_variant.sh <out-subdir>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-ranges-compare --no-ibl --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "profile run failed"; tail -5 $OUT/bench.err; exit 1; }
f=$(ls -S $(find $OUT/trace -name "*kernel_stats.csv") | head -1)
grep -E "light_classes|shade_kernel<2, true, false, true>" $f | cut -d, -f1-4
