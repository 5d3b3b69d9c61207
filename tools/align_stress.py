"""tools/align_stress.py [cases] [seed] -- randomised check of the complete-alignment kernel (k_simbits_screen_mfma<.,2>: every
pair's rmsd and max deviation after the optimal rotation) against the one-pair-per-lane exact kernel (k_pairs_exact, the
kernel the parity tests hold against the oracle) on ensembles the benchmark does not contain: 2 - 3 000 conformers, 3 - 130
atoms, planar and collinear structures, exact duplicates, mirror images, large offsets, structures of very different size.
All pairs of small ensembles, 20 000 sampled pairs otherwise (fc_ensemble_rmsd_and_max_all, symmetric output); tolerance 2e-9 absolute on both values."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
fc.init(0)
worst = 0.0
for case in range(cases):
    n = int(rng.choice([2, 3, 17, 63, 64, 65, 127, 129, 255, 300, 511, 777, 1024, 1500, 2049, 3000]))
    a = int(rng.choice([3, 4, 5, 8, 11, 16, 29, 50, 51, 64, 80, 99, 128, 130]))
    kind = str(rng.choice(["clusters", "continuous", "random", "planar", "collinear", "mirror", "duplicates", "scales"]))
    if kind == "clusters":
        X = syn.synthetic_ensemble(max(n, 2), a, seed=int(rng.integers(1 << 30)))[0][:n]
    elif kind == "continuous":
        X = syn.continuous_ensemble(n, a, seed=int(rng.integers(1 << 30)))
    else:
        X = rng.normal(scale=2.0, size=(n, a, 3))
        if kind == "planar":
            X[:, :, 2] = 0.0
        elif kind == "collinear":
            X[:, :, 1:] = 0.0
        elif kind == "mirror":
            X[1::2] = X[0::2][: len(X[1::2])] * np.array([1.0, 1.0, -1.0])
        elif kind == "duplicates":
            X[n // 2:] = X[: n - n // 2]
        elif kind == "scales":
            X *= rng.uniform(0.05, 30.0, size=(n, 1, 1))
    X = X + rng.normal(scale=float(rng.choice([0.0, 5.0, 300.0])), size=(n, 1, 3))
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, M, _ = ens.rmsd_and_max_all()   # (when every pair is degenerate -- planar, collinear -- and the fix-up queue
        if n <= 600:                       #  overflows, the entry redoes the matrix with the plain kernel)
            pi_, pj_ = np.triu_indices(n, 1)
        else:
            pi_ = rng.integers(0, n - 1, size=20000)
            pj_ = rng.integers(pi_ + 1, n)
        r, m = R[pi_, pj_], M[pi_, pj_]
        assert np.array_equal(R, R.T) and np.array_equal(M, M.T) and not R.diagonal().any()
        r0, m0 = ens.rmsd_pairs(pi_, pj_)
    scale = max(1.0, float(np.abs(r0).max()))
    err = float(max(np.abs(r - r0).max(), np.abs(m - m0).max())) / scale
    worst = max(worst, err)
    ok = bool(np.isfinite(r).all() and np.isfinite(m).all() and err < 2e-9)
    print(json.dumps({"case": case, "n": n, "atoms": a, "kind": kind, "pairs": int(len(r)), "max_err_rel_to_largest_rmsd": err,
                      "ok": ok}), flush=True)
    if not ok:
        bad = int(np.argmax(np.maximum(np.abs(r - r0), np.abs(m - m0))))
        print(json.dumps({"pair": [int(pi_[bad]), int(pj_[bad])], "got": [float(r[bad]), float(m[bad])],
                          "exact_kernel": [float(r0[bad]), float(m0[bad])]}))
        sys.exit(1)
print(json.dumps({"cases": cases, "worst": worst, "all_within": 2e-9}))
