"""Workgroup timeline of the screen kernel (needs a -DFC_TIMELINE build): occupancy over
time, block durations, fill share, ramp and tail.  Tuning tool, not part of the product."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ["FC_TIMELINE_OUT"] = out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.bin"
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

fc.init(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
X, atoms, asg = syn.synthetic_ensemble(n, 50, seed=2)
with fc.DeviceEnsemble(X, center=True) as ens:
    for _ in range(3):
        ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
start, filled, end, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
ok = end > 0
t0 = start[ok].min()
us = lambda x: (x - t0) / 100.0  # 100 MHz wall clock
s, f, e = us(start[ok]), us(filled[ok]), us(end[ok])
dur = e - s
print("blocks", ok.sum(), "of", len(t), "span us", e.max(), "sum dur us", dur.sum(), "mean active", dur.sum() / e.max())
print("duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("fill us: mean %.2f p50 %.2f p90 %.2f max %.2f ; share of block time %.3f" % ((f - s).mean(), *np.percentile(f - s, [50, 90]), (f - s).max(), (f - s).sum() / dur.sum()))
edges = np.linspace(0, e.max(), 41)
act = [((s < b) & (e > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("active blocks per 2.5% slice:", act)
print("last start us %.1f, ends after last start: %d blocks; time from 95%% of blocks ended to end: %.1f us" % (s.max(), (e > s.max()).sum(), e.max() - np.percentile(e, 95)))
wg = hw[ok]  # persistent workgroup that ran the item
order = np.argsort(s)
gaps = []
for g in np.unique(wg):
    sel = order[wg[order] == g]
    gaps.extend((s[sel][1:] - e[sel][:-1]).tolist())
gaps = np.array(gaps)
print("workgroups %d, items per workgroup %.1f, gap between consecutive items of a workgroup: mean %.2f us p90 %.2f us" % (len(np.unique(wg)), len(s) / len(np.unique(wg)), gaps.mean() if len(gaps) else 0, np.percentile(gaps, 90) if len(gaps) else 0))
