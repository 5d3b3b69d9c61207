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
for rep in range(2):
    t0 = time.perf_counter()
    tf, rot = fc.torsion_module.torsion_scan_fingerprints(base, torsions, masks, angles, torsions, thresh=1.5)
    t1 = time.perf_counter()
    kept = np.flatnonzero(rot != 0)
    tf_all = np.concatenate([fc.torsion_module.get_torsion_fingerprint(base, torsions)[None], tf[kept]])
    t2 = time.perf_counter()
    N, Q = tf_all.shape
    fm = np.zeros(N, dtype=np.int64)
    L.call("fc_tfd_first_match", L.pf(tf_all), N, Q, 10.0, L.pi(fm))
    t3 = time.perf_counter()
    mask = np.zeros(N, dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    t4 = time.perf_counter()
    print(json.dumps({"scan": t1 - t0, "host_concat": t2 - t1, "first_match_call": t3 - t2, "ladder_call": t4 - t3, "kept": int(mask.sum()), "matched": int((fm >= 0).sum())}))
d = (fm - np.arange(N))[fm >= 0]
print(json.dumps({"first_match_distance_quantiles": {str(q): int(np.quantile(d, q)) for q in (0.1, 0.5, 0.9, 0.99, 0.999, 1.0)},
                  "mean_distance": float(d.mean()), "sum_distance": float(d.sum()), "unmatched": int((fm < 0).sum()),
                  "unmatched_sum_remaining": float((N - np.flatnonzero(fm < 0)).sum()),
                  "per_block_max_distance_sum": float(np.where(fm >= 0, fm - np.arange(N), N - np.arange(N))[: (N // 64) * 64].reshape(-1, 64).max(axis=1).sum())}))
