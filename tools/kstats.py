"""tools/kstats.py <kernel_stats.csv> [substring ...] -- calls and mean duration (us) of the kernels of a rocprofv3 --stats
summary whose names contain one of the substrings (all kernels without one), mangled template names cut short"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    name = r["Name"]
    if pats and not any(p in name for p in pats):
        continue
    short = name.split("(")[0][:60]
    print("%-62s calls %5d  mean %9.2f us  total %9.3f ms" % (short, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
