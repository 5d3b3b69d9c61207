"""tools/trace_phases.py KERNEL_TRACE.csv MARKER -- the LAST run of a rocprofv3 kernel trace (from the last dispatch whose name
contains MARKER) as a timeline: consecutive dispatches grouped while their names stay in one family, with start / end
relative to the run's start (ms), the number of kernels, the time the device was busy inside the group and the idle gap
in front of it."""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
s0 = [s for s, e, n in ev if marker in n][-1]
last = [x for x in ev if x[0] >= s0]


def family(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    for key, fam in (("k_ts_", "scan tree"), ("k_torsion_scan", "scan rows"), ("first_match", "first match"), ("tfd_pack", "first match"),
                     ("window_bounds", "first match"), ("k_chunk", "ladder"), ("k_comp", "ladder"), ("k_c_", "ladder"), ("k_tiny", "ladder"),
                     ("k_apply", "ladder"), ("k_fill_u8", "ladder"), ("LadTab", "ladder"), ("simbits", "rmsd prune"), ("refine", "rmsd prune"),
                     ("k_ladder", "rmsd prune"), ("k_pair", "rmsd prune"), ("rocprim", "scan / sort"), ("rocclr", "copy / fill")):
        if key in n:
            return fam
    return n[-28:]


groups = []
for s, e, n in last:
    f = family(n)
    if f in ("scan / sort", "copy / fill") and groups:
        f = groups[-1][0]
    if groups and groups[-1][0] == f:
        g = groups[-1]
        g[2] = max(g[2], e)
        g[3] += 1
        g[4].append((s, e))
    else:
        groups.append([f, s, e, 1, [(s, e)]])
gaps = []
pe = s0
for s_, e_, n_ in last:
    if s_ - pe > 50000:
        gaps.append(((s_ - pe) / 1e6, (pe - s0) / 1e6, n_.replace("(anonymous namespace)::", "").split("(")[0][-40:]))
    pe = max(pe, e_)
prev_end = s0
for f, s, e, k, iv in groups:
    busy, cs, ce = 0, None, None
    for a, b in sorted(iv):
        if ce is None or a > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    print("%8.3f -> %8.3f ms  %-14s %4d kernels  busy %7.3f ms  gap before %6.3f ms" % ((s - s0) / 1e6, (e - s0) / 1e6, f, k, busy / 1e6, (s - prev_end) / 1e6))
    prev_end = max(prev_end, e)
print("idle gaps of more than 0.05 ms (length, at, next kernel):")
for g in gaps:
    print("  %6.3f ms at %7.3f ms before %s" % g)
