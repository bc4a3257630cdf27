#!/bin/bash
# lane utilisation of the point-shadow kernel (scratch/shadow_bench.py)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2/pmc_shadow2
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -- python3 $R/scratch/shadow_bench.py > $OUT/a.log 2>&1 || { echo failed; tail -5 $OUT/a.log; exit 1; }
python3 - <<'P'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2/pmc_shadow2"
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "point_shadow" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    d = rows[k]
    util = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64) if d.get("SQ_ACTIVE_INST_VALU") else 0
    print(k, {n: f"{v:.3g}" for n, v in d.items()}, f"lane utilisation {util:.3f}")
P
