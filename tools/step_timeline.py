"""Timeline of a few steady-state prune steps from a rocprofv3 kernel trace CSV (one line per dispatch: start, end,
duration, queue, kernel).  Usage: python tools/step_timeline.py <kernel_trace.csv> [first_step] [n_steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [k for k, r in enumerate(rows) if "screen_mfma_h2" in r["Kernel_Name"]]
a, b = idx[first], idx[first + n]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-36:]
    print("%8.1f -> %8.1f (%6.1f) q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), name))
starts = [int(rows[k]["Start_Timestamp"]) for k in idx]
ends = [int(rows[k]["End_Timestamp"]) for k in idx]
gaps = [(starts[k + 1] - ends[k]) / 1e3 for k in range(10, len(idx) - 1)]
durs = [(ends[k] - starts[k]) / 1e3 for k in range(10, len(idx))]
print("screens: %d, mean duration %.1f us, mean gap to the next screen %.1f us (min %.1f, max %.1f)" %
      (len(idx), sum(durs) / len(durs), sum(gaps) / len(gaps), min(gaps), max(gaps)))
