"""Workgroup timeline of the split-half screen (needs a -DFC_H2_TIMELINE build: make BUILD=build_h2tl
OUT=../libfc_hip_h2tl.so EXTRA=-DFC_H2_TIMELINE; FC_LIB_PATH selects it): workgroup durations, share of the
column-tile fill, gap between two workgroups on a CU slot, ramp and tail.  Tuning tool, not part of the product."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ["FC_H2_TIMELINE_OUT"] = out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/h2_timeline.bin"
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

fc.init(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
a = int(sys.argv[3]) if len(sys.argv) > 3 else 50
X, atoms, asg = syn.synthetic_ensemble(n, a, seed=2)
with fc.DeviceEnsemble(X, center=True) as ens:
    for _ in range(3):
        ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
start, filled, end, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64), t[:, 3]
ok = end > 0
t0 = start[ok].min()
us = lambda x: (x - t0) / 100.0  # 100 MHz wall clock
s, f, e = us(start[ok]), us(filled[ok]), us(end[ok])
dur = e - s
print("workgroups", ok.sum(), "of", len(t), "span us %.1f" % e.max(), "sum of durations us %.0f" % dur.sum(), "mean resident %.1f" % (dur.sum() / e.max()))
print("duration us: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("fill us: mean %.2f p50 %.2f p90 %.2f max %.2f ; share of workgroup time %.3f" % ((f - s).mean(), *np.percentile(f - s, [50, 90]), (f - s).max(), (f - s).sum() / dur.sum()))
edges = np.linspace(0, e.max(), 41)
print("resident workgroups per 2.5% slice:", [int(((s < b) & (e > a_)).sum()) for a_, b in zip(edges[:-1], edges[1:])])
print("last start us %.1f; time from 95%% of the workgroups ended to the end: %.1f us" % (s.max(), e.max() - np.percentile(e, 95)))
# CU = (xcc, se, sh, cu); a CU holds up to 3 of these workgroups: gaps between an end and the next start on the same CU
hwv = hw[ok]
hwid, xcc = (hwv & 0xffffffff).astype(np.int64), (hwv >> 32).astype(np.int64) & 0xf
cu = (xcc << 12) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
print("distinct CUs seen:", len(np.unique(cu)))
gaps, conc = [], []
for c in np.unique(cu):
    sel = np.flatnonzero(cu == c)
    o = sel[np.argsort(s[sel])]
    ends = np.sort(e[o])
    # for every start (after the first three) the time since the most recent end on this CU before it
    for k in o[3:]:
        prev = ends[ends <= s[k] + 1e-9]
        if len(prev):
            gaps.append(s[k] - prev[-1])
    conc.append(dur[sel].sum() / (e[sel].max() - s[sel].min()))
gaps = np.array(gaps)
print("start minus latest earlier end on the same CU: mean %.2f us p50 %.2f p90 %.2f" % (gaps.mean(), *np.percentile(gaps, [50, 90])))
print("mean resident workgroups per CU over its busy span: %.2f" % np.mean(conc))
