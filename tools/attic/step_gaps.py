"""Idle time between the kernels of one prune step, from a rocprofv3 kernel trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows]
# steps start at each screen kernel <4, false>
idx = [k for k, e in enumerate(ev) if "screen_mfma<4, false>" in rows[k]["Kernel_Name"]]
for a, b in list(zip(idx, idx[1:]))[3:6]:
    t0 = ev[a][0]
    print("--- step")
    prev_end = None
    for s, e, n in ev[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("%9.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
        prev_end = e
    print("next step starts %.1f us after this one's last kernel ended" % ((ev[b][0] - prev_end) / 1e3))
