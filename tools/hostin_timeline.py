"""Device-side timeline of ONE prune_by_rmsd(host arrays) call at BASELINE configs[1] (10 000 x 50).

  run:     rocprofv3 --kernel-trace --memory-copy-trace -d DIR --output-format csv -- python3 tools/hostin_timeline.py
           (prints the host-side wall time of each call as JSON; FC_TIMELINE_N / FC_TIMELINE_A: another shape,
           FC_TIMELINE_PINNED=1: the ensemble in page-locked memory)
  report:  python3 tools/hostin_timeline.py DIR      (kernels and copies of the LAST call, microseconds from its first one)
"""
import csv
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import firecode_amd as fc
    from firecode_amd import synthetic as syn

    fc.init(0)
    n_conf, n_atoms = int(os.environ.get("FC_TIMELINE_N", "10000")), int(os.environ.get("FC_TIMELINE_A", "50"))
    X, atoms, _ = syn.synthetic_ensemble(n_conf, n_atoms, seed=2)
    pinned = os.environ.get("FC_TIMELINE_PINNED") == "1"
    if pinned:
        P = fc.pinned_empty(X.shape)
        P[...] = X
        X = P
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        fc.pruner.prune_by_rmsd(X, atoms, 0.5)
        ts.append((time.perf_counter() - t0) * 1e3)
        time.sleep(0.02)  # a gap in the trace between the calls
    print(json.dumps({"pinned_source": pinned, "call_ms": [round(t, 4) for t in ts], "min_ms": round(min(ts), 4)}))


def report(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                       "copy %s %s B" % (r.get("Direction", "?"), r.get("Size", "?"))))
    ev.sort()
    # calls are separated by the 20 ms sleeps: split at gaps above 5 ms, report the last group
    groups, cur = [], []
    for e in ev:
        if cur and e[0] - max(x[1] for x in cur) > 5_000_000:
            groups.append(cur)
            cur = []
        cur.append(e)
    if cur:
        groups.append(cur)
    g = groups[-1]
    t0 = g[0][0]
    busy, last_end = 0, t0
    for s, e, name in g:
        print("%8.1f -> %8.1f (%6.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name))
        busy += max(0, e - max(s, last_end))
        last_end = max(last_end, e)
    span = (last_end - t0) / 1e3
    print("events %d, span %.1f us, device or copy engine busy %.1f us (%.2f)" % (len(g), span, busy / 1e3, busy / 1e3 / span))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        report(sys.argv[1])
    else:
        run()
