#!/bin/bash
# counters of the by-region irradiance kernel (scratch/irr_rows_bench.py): lanes, texture addresser, L2
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/pmc_irr_rows
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/scratch/irr_rows_bench.py > $OUT/a.log 2>&1 || { echo failed a; tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/scratch/irr_rows_bench.py > $OUT/b.log 2>&1 || { echo failed b; tail -5 $OUT/b.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d $OUT/c -- python3 $R/scratch/irr_rows_bench.py > $OUT/c.log 2>&1 || { echo failed c; tail -5 $OUT/c.log; exit 1; }
python3 - <<'P'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3/pmc_irr_rows"
for p in "abc":
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "irradiance" in r["Kernel_Name"]:
                rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(rows)[-3:]:
        print(k, {n: f"{v:.4g}" for n, v in rows[k].items()})
P
