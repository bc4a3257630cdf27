#!/bin/bash
# Instruction counters of light_classes_kernel with the library in VKR_HIP_LIB.  usage: bash scratch/pmc_classes.sh <out-subdir>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py --no-ranges-compare --no-ibl --no-coverage --cpu-rows 0 --steps 3 --warmup 1 > $OUT/sq.log 2>&1 || { echo failed; tail -5 $OUT/sq.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = max(glob.glob(sys.argv[1] + "/sq/*/*counter_collection.csv"), key=lambda p: sum(1 for _ in open(p)))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "light_classes_kernel" in r["Kernel_Name"] and r["Grid_Size"] not in ("",):
        acc[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k, len(v), f"{sum(v)/len(v):.6g}")
PY
