"""tools/kstats_ladder.py <dir> [calls_per_unit]: kernel_stats.csv of a rocprofv3 run, names cut to the kernel, time per unit"""
import csv, re, glob, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_stats.csv")[0]
per = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    n = r["Name"]
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    short = (m.group(0) if m else n[:50])
    tot += float(r["TotalDurationNs"])
    print("%-40s calls %4d mean %9.2f us per-unit %8.3f ms" % (short[:40], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / per / 1e6))
print("sum per unit ms %.3f, launches per unit %.1f" % (tot / per / 1e6, sum(int(r["Calls"]) for r in rows) / per))
