"""Why the re-scan of the survivors of a cfg3 search sometimes takes 25 ms instead of 1: times two identical torsion_scan calls
behind every fc_torsion_scan_tfd_grid call."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
fc._lib.warmup()
rng = np.random.default_rng(3)
A, T = 50, 8
base = syn.synthetic_skeleton(A, rng)
centres = np.linspace(3, A - 6, T).astype(int)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
masks = np.zeros((T, A), dtype=bool)
for t, c in enumerate(centres):
    masks[t, c + 2:] = True
values = [(0, 60, 120, 180, 240, 300)] * T
for run in range(6):
    t0 = time.perf_counter()
    rot, keep = fc.torsion_module.torsion_scan_tfd_grid(base, torsions, masks, values, torsions, thresh=1.5, tfd_thresh=10)
    t1 = time.perf_counter()
    ang = fc.utils.cartesian_rows_at(values, np.flatnonzero(keep[1:]))
    t2 = time.perf_counter()
    a = fc.torsion_module.torsion_scan(base, torsions, masks, ang, thresh=1.5)[0]
    t3 = time.perf_counter()
    b = fc.torsion_module.torsion_scan(base, torsions, masks, ang, thresh=1.5)[0]
    t4 = time.perf_counter()
    print(json.dumps({"scan_tfd": round(t1 - t0, 4), "rows_at": round(t2 - t1, 4), "rescan_1": round(t3 - t2, 4), "rescan_2": round(t4 - t3, 4)}))

# is it the library at all?  first touch of a fresh 4.6 MB NumPy array in this process, behind the same call
for run in range(6):
    rot, keep = fc.torsion_module.torsion_scan_tfd_grid(base, torsions, masks, values, torsions, thresh=1.5, tfd_thresh=10)
    t0 = time.perf_counter()
    x = np.empty((3859, 50, 3))
    x.fill(1.0)
    t1 = time.perf_counter()
    y = np.empty((3859, 50, 3))
    y.fill(1.0)
    t2 = time.perf_counter()
    print(json.dumps({"first_touch_4.6MB_ms": round(1e3 * (t1 - t0), 2), "again_ms": round(1e3 * (t2 - t1), 2)}))
    del x, y
