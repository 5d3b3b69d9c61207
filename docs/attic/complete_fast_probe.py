"""tools/complete_fast_probe.py -- the complete alignments' single-precision atom pass against the fp64 pass:
values of whole (N, N) matrices on several kinds of ensembles (clusters, continuous, exact and near duplicates, symmetric
structures whose atoms tie for the largest deviation, far from the origin), then the time of both forms at 10 000 x 50.
Usage: python tools/complete_fast_probe.py [values|time|all]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import _lib as L  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
fc.init(0)


def both(X):
    out = []
    for allow in (True, False):
        L.complete_select(allow)
        with fc.DeviceEnsemble(X, center=True) as ens:
            r, m, ms = ens.rmsd_and_max_all()
            _, _, st = ens.bench_rmsd_and_max_all(1)
        out.append((r, m, L.complete_last_form(), int(st[1])))
    L.complete_select(True)
    return out


def ensembles():
    rng = np.random.default_rng(7)
    yield "clusters 2000 x 50", syn.synthetic_ensemble(2000, 50, seed=3)[0]
    yield "continuous 1500 x 50", syn.continuous_ensemble(1500, 50, seed=4)
    yield "continuous 1200 x 23", syn.continuous_ensemble(1200, 23, seed=5)
    yield "continuous 1000 x 64", syn.continuous_ensemble(1000, 64, seed=6)
    X = syn.continuous_ensemble(900, 40, seed=8)
    X[300:600] = X[:300]  # exact duplicates
    X[600:] = X[:300] + rng.normal(scale=1e-5, size=(300, 40, 3))  # and near ones
    yield "duplicates 900 x 40", X
    # atoms that tie for the largest deviation: a structure with a mirror plane, its copies displaced symmetrically
    base = rng.normal(scale=2.0, size=(12, 3))
    sym = np.concatenate([base, base * np.array([1.0, 1.0, -1.0])])  # 24 atoms, mirror z -> -z
    X = np.repeat(sym[None], 600, axis=0).copy()
    amp = rng.normal(scale=0.3, size=(600, 12, 3))
    X[:, :12] += amp
    X[:, 12:] += amp * np.array([1.0, 1.0, -1.0])
    yield "mirror-symmetric 600 x 24", X
    yield "far from the origin 800 x 30", syn.continuous_ensemble(800, 30, seed=9) + np.array([250.0, -90.0, 40.0])
    yield "large coordinates 700 x 32", syn.continuous_ensemble(700, 32, seed=10) * 40.0


if what in ("values", "all"):
    for name, X in ensembles():
        (r1, m1, f1, q1), (r0, m0, f0, q0) = both(X)
        iu = np.triu_indices(len(X), 1)
        dr, dm = np.abs(r1 - r0)[iu], np.abs(m1 - m0)[iu]
        print(json.dumps({"ensemble": name, "forms": [f1, f0], "fixup_pairs": [q1, q0], "max_abs_rmsd_diff": float(dr.max()),
                          "max_abs_maxdev_diff": float(dm.max()), "maxdev_bit_equal": float((m1[iu] == m0[iu]).mean()),
                          "rmsd_range": [float(r0[iu].min()), float(r0[iu].max())]}), flush=True)

if what in ("time", "all"):
    for n, a in ((10000, 50), (8000, 32), (8000, 64)):
        X = syn.synthetic_ensemble(n, a, seed=2)[0]
        rec = {"n": n, "a": a}
        for allow in (True, False, True, False):
            L.complete_select(allow)
            with fc.DeviceEnsemble(X, center=True) as ens:
                ens.bench_rmsd_and_max_all(2)
                k, t, st = ens.bench_rmsd_and_max_all(10)
            pairs = n * (n - 1) // 2
            rec.setdefault("form_%d" % L.complete_last_form(), []).append(
                {"kernel_ms": round(k, 4), "frac_fp64_peak": round(pairs * (53 * a + 600) / (k * 1e-3) / 78.6e12, 4), "fixup_pairs": int(st[1])})
        L.complete_select(True)
        print(json.dumps(rec), flush=True)
