"""cfg3's first-match step alone: the fingerprints of the systematic scan (1 679 611 x 8), `fc_tfd_first_match` timed
over a few repeats, the result compared with a kept array (tools/cfg3_fm.npz, from tools/dump_cfg3_fm.py).
  python tools/fm_probe.py [dump]     dump: also write the fingerprints (compressed) to gpurun_out/cfg3_tf.npz"""
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
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg3_fm.npz")
if os.path.exists(p):
    ref = np.load(p)["fm"].astype(np.int64)
fm = np.zeros(N, dtype=np.int64)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    L.call("fc_tfd_first_match", L.pf(tf_all), N, Q, 10.0, L.pi(fm))
    ts.append(time.perf_counter() - t0)
print(json.dumps({"N": int(N), "Q": int(Q), "call_ms": [round(1e3 * t, 3) for t in ts],
                  "same_as_kept": None if ref is None or len(ref) != N else bool(np.array_equal(ref, fm)),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("FC_TFD")}}))
if len(sys.argv) > 1 and sys.argv[1] == "dump":
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/cfg3_tf.npz", tf=tf_all)
