"""tools/ts_stamps.py -- tuning build only (make BUILD=build_stamps OUT=../libfc_hip_stamps.so EXTRA="-DFC_TUNING_BUILD -DFC_TFD_STAMPS",
FC_LIB_PATH=firecode_amd/libfc_hip_stamps.so): counters of the scan tree's last level on cfg3 (nodes, clashes, back-off steps, ticks)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L, synthetic as syn
fc.init(0)
lib = L.load()
rng = np.random.default_rng(3)
A, T = 50, 8
base = syn.synthetic_skeleton(A, rng)
centres = np.linspace(3, A - 6, T).astype(int)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
masks = np.zeros((T, A), dtype=bool)
for t, c in enumerate(centres):
    masks[t, c + 2:] = True
angles = fc.utils.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * T)
out = (C.c_ulonglong * 16)()
for rep in range(2):
    tf, rot = fc.torsion_module.torsion_scan_fingerprints(base, torsions, masks, angles, torsions, thresh=1.5)
    assert lib.fc_debug_ts_stamps(out, 1) == 0
    v = [int(x) for x in out]
    n = max(v[0], 1)
    print({"nodes": v[0], "angle!=0": v[1], "nodes into back-off (all levels)": v[2], "back-off steps (all levels)": v[3],
           "us per node: state to LDS": v[4] / n / 100, "torsion step": v[5] / n / 100, "outputs": v[6] / n / 100,
           "closed form refused": v[7], "waves": v[9], "us per wave": v[8] / max(v[9], 1) / 100})
