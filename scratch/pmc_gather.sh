#!/bin/bash
# texture-addresser cycles per gather instruction for the patterns of scratch/ubench/gather_bench (one dispatch per pattern, mode and repeat)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/pmc_gather
mkdir -p $OUT
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/a -- $R/scratch/ubench/gather_bench > $OUT/a.log 2>&1 || { echo failed; tail -5 $OUT/a.log; exit 1; }
cat $OUT/a.log
python3 - <<'P'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3/pmc_gather"
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        rows[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"][:40]
pats = ["coalesced row", "8x8 16K", "4x16 16K", "8x8 4M", "random 16K", "random 2M", "random 64M", "random 256M"]
ids = sorted(rows)
for n, d in enumerate(ids):
    c = rows[d]
    pat, mode, rep = n // 10, (n % 10) // 2, n % 2
    if rep == 0 or "SQ_INSTS_VMEM_RD" not in c: continue
    g = c["SQ_INSTS_VMEM_RD"]
    print(f"{pats[pat] if pat < len(pats) else pat:14s} mode {mode}: TA busy cycles per gather {c['TA_TA_BUSY_sum'] / g:6.2f}  (TA busy {c['TA_TA_BUSY_sum'] / 256 / (c['GRBM_GUI_ACTIVE'] / 8):.2f} of the time), L1 accesses per gather {c['TCP_TOTAL_CACHE_ACCESSES_sum'] / g:5.1f}")
P
