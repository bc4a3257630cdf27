#!/bin/bash
# Counters of the point-shadow kernel (scratch/point_shadow_time.py).  usage: VKR_HIP_LIB=... bash scratch/pmc_point_shadow.sh <out-subdir>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/scratch/point_shadow_time.py > $OUT/sq.log 2>&1 || { echo failed; tail -5 $OUT/sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $OUT/tc -- python3 $R/scratch/point_shadow_time.py > $OUT/tc.log 2>&1 || { echo failed tc; tail -5 $OUT/tc.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for sub in ("sq", "tc"):
    f = glob.glob(sys.argv[1] + f"/{sub}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "point_shadow_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:40s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
