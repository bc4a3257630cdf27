#!/bin/bash
# issue-side counters of the prefilter launches of scratch/ibl_bench.py
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2/pmc_prefilter
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/scratch/ibl_bench.py > $OUT/a.log 2>&1 || { echo failed a; tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $R/scratch/ibl_bench.py > $OUT/b.log 2>&1 || { echo failed b; tail -5 $OUT/b.log; exit 1; }
python3 - <<'P'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2/pmc_prefilter"
for p in "ab":
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "prefilter" in r["Kernel_Name"] and "ILb1" in r["Kernel_Name"] or "prefilter_texel_kernel<true>" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(rows)[-5:]:
        print(p, k, {n: f"{v:.4g}" for n, v in rows[k].items()})
P
