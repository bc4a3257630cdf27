"""Prints the counters of one kernel from the CSVs that scratch/pmc_irr_rows.sh leaves (argument: substring of the kernel name)."""
import csv, glob, collections, sys
out = "gpurun_out/r3/pmc_irr_rows"
name = sys.argv[1] if len(sys.argv) > 1 else "irradiance_rows"
for p in "abc":
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(rows)[-1:]:
        print(k, {n: f"{v:.4g}" for n, v in rows[k].items()})
