#!/bin/bash
# Instruction counters of the shading kernel with and without depth ranges (scratch/ranges_one.py launches both forms).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE TA_TA_BUSY_sum --output-format csv -d $OUT/sq -- python3 $R/scratch/ranges_one.py > $OUT/sq.log 2>&1 || { echo failed; tail -5 $OUT/sq.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/sq/*/*counter_collection.csv")[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "shade_kernel" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for d, v in sorted(by.items()):
    key = round(v["SQ_INSTS_VMEM_RD"])
    if key not in seen:
        seen[key] = v
        print(d, {k: f"{x:.5g}" for k, x in sorted(v.items())})
PY
