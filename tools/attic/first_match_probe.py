import sys, time, json, os
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
tf_all = np.ascontiguousarray(np.concatenate([fc.torsion_module.get_torsion_fingerprint(base, torsions)[None], tf[kept]]))
N, Q = tf_all.shape
ref = None
for look in ("0", "4096", "32768", "131072", "300000"):
    os.environ["FC_TFD_LOOKAHEAD"] = look
    fm = np.zeros(N, dtype=np.int64)
    L.call("fc_tfd_first_match", L.pf(tf_all), N, Q, 10.0, L.pi(fm))
    t0 = time.perf_counter()
    L.call("fc_tfd_first_match", L.pf(tf_all), N, Q, 10.0, L.pi(fm))
    dt = time.perf_counter() - t0
    if ref is None: ref = fm.copy()
    print(json.dumps({"lookahead": look, "call_s": dt, "equal_to_one_phase": bool(np.array_equal(fm, ref))}))
# window statistics: how wide are the boxes
tfF = tf_all[:, :4].astype(np.float32)
nw = N // 1024
lo = tfF[:nw*1024].reshape(nw, 1024, 4).min(1); hi = tfF[:nw*1024].reshape(nw, 1024, 4).max(1)
print(json.dumps({"box_width_quantiles_per_component": [[float(np.quantile(hi[:,q]-lo[:,q], x)) for x in (0.1,0.5,0.9)] for q in range(4)]}))
