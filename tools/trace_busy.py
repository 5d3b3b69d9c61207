"""tools/trace_busy.py KERNEL_TRACE.csv MARKER -- device-busy share of the LAST run in a rocprofv3 kernel trace: the run starts at
the last dispatch whose kernel name contains MARKER; busy = union of the kernels' [start, end) intervals."""
import csv
import json
import sys

path, marker = sys.argv[1], sys.argv[2]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
starts = [s for s, e, n in ev if marker in n]
s0 = starts[-1]
last = [x for x in ev if x[0] >= s0]
end = max(e for _, e, _ in last)
busy, cs, ce = 0, None, None
for s, e, _ in last:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
per = {}
for s, e, n in last:
    k = n.replace("(anonymous namespace)::", "").split("(")[0][-48:]
    per[k] = per.get(k, 0) + (e - s)
top = sorted(per.items(), key=lambda kv: -kv[1])[:12]
print(json.dumps({"runs_in_trace": len(starts), "last_run_span_ms": (end - s0) / 1e6, "device_busy_ms": busy / 1e6,
                  "device_busy_share": busy / (end - s0), "kernels_in_last_run": len(last),
                  "kernel_time_summed_over_streams_ms": sum(per.values()) / 1e6,
                  "largest_kernels_ms": {k: v / 1e6 for k, v in top}}, indent=1))
