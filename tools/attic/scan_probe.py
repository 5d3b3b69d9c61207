import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
rng = np.random.default_rng(3)
A, T = 50, 8
base = syn.synthetic_skeleton(A, rng)
centres = np.linspace(3, A - 6, T).astype(int)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
masks = np.zeros((T, A), dtype=bool)
for t, c in enumerate(centres):
    masks[t, c + 2:] = True
angles = fc.utils.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * T)
for thresh in (1.5, 1.5, 0.0, 0.0, 1.0, 2.5):
    t0 = time.perf_counter()
    tf, rot = fc.torsion_module.torsion_scan_fingerprints(base, torsions, masks, angles, torsions, thresh=thresh)
    t1 = time.perf_counter()
    print(json.dumps({"thresh": thresh, "s": round(t1 - t0, 4), "rot_hist": np.bincount(rot, minlength=9).tolist()}))
