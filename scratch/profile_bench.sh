#!/bin/bash
# Kernel-trace statistics of the default benchmark command.  usage: bash scratch/profile_bench.sh <out-subdir>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "profile run failed"; tail -5 $OUT/bench.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo done
