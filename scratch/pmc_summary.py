"""Summarises rocprofv3 --pmc CSVs: per kernel name, mean of each counter over dispatches."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "shade_kernel"
acc = collections.defaultdict(list)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
