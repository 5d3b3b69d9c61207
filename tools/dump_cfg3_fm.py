"""Dump the cfg3 first-match array (and the ladder's mask) so that the ladder can be studied on a CPU box."""
import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L, synthetic as syn
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
tf, rot = fc.torsion_module.torsion_scan_fingerprints(base, torsions, masks, angles, torsions, thresh=1.5)
kept = np.flatnonzero(rot != 0)
tf_all = np.concatenate([fc.torsion_module.get_torsion_fingerprint(base, torsions)[None], tf[kept]])
N, Q = tf_all.shape
fm = np.zeros(N, dtype=np.int64)
L.call("fc_tfd_first_match", L.pf(tf_all), N, Q, 10.0, L.pi(fm))
for rep in range(3):
    mask = np.zeros(N, dtype=np.uint8)
    t0 = time.perf_counter()
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    print(json.dumps({"ladder_call_s": time.perf_counter() - t0, "kept": int(mask.sum()), "N": int(N)}))
np.savez_compressed("gpurun_out/cfg3_fm.npz", fm=fm.astype(np.int32), mask=np.packbits(mask))
