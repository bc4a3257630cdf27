"""kstats.py DIR SUBSTRING: average duration per kernel whose name holds SUBSTRING, from rocprofv3 --stats output under DIR."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Name"]:
            print(f"   {row['Name'].split('(')[0][:60]:60s} calls {row['Calls']:>4s}  avg {float(row['AverageNs']) / 1e6:.3f} ms")
