"""tools/ladder_timeline.py <dir>: start / end of every kernel of the LAST ladder call in a rocprofv3 kernel trace (ms from its first kernel)"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last k_fill_u8
starts = [i for i, r in enumerate(rows) if "k_fill_u8" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
agg = {}
for r in rows:
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"][:40]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    a = agg.setdefault(name, [s, e, 0, 0.0])
    a[0] = min(a[0], s); a[1] = max(a[1], e); a[2] += 1; a[3] += e - s
for name, (s, e, n, busy) in sorted(agg.items(), key=lambda kv: kv[1][0]):
    print("%-40s x%3d  %7.3f .. %7.3f ms  (busy %6.3f)" % (name[:40], n, s, e, busy))
