"""tools/refine_stress.py [cases] [seed] -- randomised equality check of the two long-queue refine forms and the two ladder
forms: for every case the FIRST prune of a fresh ensemble takes the straight queue walk (k_refine_pairs) and the one-workgroup
ladder, the SECOND one (the library has seen the queue by then) the bucket refine and, for long lists, the per-level ladder;
similarity bits, counts and masks must be identical, and a pipelined batch of prunes must give the same mask again.
Ensembles without cluster structure, 1 500 - 6 000 conformers, 3 - 110 atoms, thresholds that give > 2^17 candidates."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fc.init(0)
done, long_q, long_l = 0, 0, 0
for case in range(cases):
    n = int(rng.integers(1500, 6000))
    a = int(rng.choice([3, 4, 7, 13, 23, 30, 31, 50, 64, 80, 104, 110]))
    thr = float(rng.uniform(0.55, 1.1))
    X = syn.continuous_ensemble(n, a, seed=int(rng.integers(1 << 30)), thr=0.5)
    if rng.random() < 0.3:  # random rigid motions on top
        X = np.einsum("nij,naj->nai", np.array([syn.random_rotation(rng) for _ in range(n)]), X) + rng.normal(scale=2.0, size=(n, 1, 3))
    with fc.DeviceEnsemble(X, center=True) as ens:
        m1, s1 = ens.prune(thr, 2 * thr)          # straight walk, one-workgroup ladder
        b1, g1 = ens.simbits(thr, 2 * thr)
        m2, s2 = ens.prune(thr, 2 * thr)          # bucket refine, per-level ladder (when the queue / list are long)
        b2, g2 = ens.simbits(thr, 2 * thr)
        _, _, m3, s3 = ens.bench_prune(thr, 2 * thr, reps=4, want_mask=True)
    ok = (np.array_equal(m1, m2) and np.array_equal(m1, m3) and np.array_equal(b1, b2) and g1 == g2 and
          list(s1[1:4]) == list(s2[1:4]) and int(s3[2]) == int(s1[2]))
    long_q += int(s1[1]) > (1 << 17)
    long_l += int(s1[2]) > (1 << 17)
    done += 1
    print(json.dumps({"case": case, "n": n, "atoms": a, "thr": round(thr, 3), "candidates": int(s1[1]), "similar": int(s1[2]),
                      "survivors": int(m1.sum()), "ok": bool(ok)}), flush=True)
    if not ok:
        sys.exit(1)
print(json.dumps({"cases": done, "with_long_queue": long_q, "with_long_similar_list": long_l, "all_equal": True}))
