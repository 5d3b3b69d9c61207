"""Two-phase first match (bounded look-ahead + column chunks with window boxes: the default from 65 536 rows on)
against the one-phase kernel on random fingerprint arrays: clusters, singletons, angles at the wrap-around,
sorted and shuffled orders.  One-off validation: python tools/first_match_stress.py [cases] [seed]"""
import json
import os
import sys

sys.path.insert(0, "/root/repo")
import numpy as np

import firecode_amd as fc
from firecode_amd import _lib as L

fc.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = []
for case in range(n_cases):
    n = int(rng.integers(70000, 400000))
    q = int(rng.choice([1, 3, 4, 6, 8, 9, 16]))
    k = int(rng.integers(50, 20000))
    centres = rng.uniform(-180, 180, size=(k, q))
    if rng.random() < 0.5:
        centres = np.round(centres / 60.0) * 60.0  # grid-like: many equal angles, values at +-180
    tf = centres[rng.integers(0, k, n)] + rng.normal(scale=rng.choice([0.01, 1.0, 3.0]), size=(n, q))
    lone = rng.integers(0, n, n // 50)
    tf[lone] = rng.uniform(-180, 180, size=(len(lone), q))
    tf = (tf + 180) % 360 - 180
    if rng.random() < 0.5:
        tf = tf[np.lexsort(tf.T[::-1])]  # sorted: long runs of equal leading angles, like a systematic scan
    tf = np.ascontiguousarray(tf)
    out = {}
    for look in ("0", None, "300"):
        if look is None:
            os.environ.pop("FC_TFD_LOOKAHEAD", None)
        else:
            os.environ["FC_TFD_LOOKAHEAD"] = look
        fm = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(tf), n, q, 10.0, L.pi(fm))
        out[look] = fm
    os.environ.pop("FC_TFD_LOOKAHEAD", None)
    ok = np.array_equal(out["0"], out[None]) and np.array_equal(out["0"], out["300"])
    print(json.dumps({"case": case, "n": n, "q": q, "unmatched": int((out["0"] < 0).sum()), "equal": bool(ok)}), flush=True)
    if not ok:
        bad.append(case)
print(json.dumps({"cases": n_cases, "mismatching_cases": bad}))
sys.exit(1 if bad else 0)
