"""tools/time_complete.py -- time the complete all-pairs alignment kernel (fc_bench_rmsd_and_max_all) on the
BASELINE configs[1] ensemble and spot-check a sample of its outputs against the oracle.
Usage: python tools/time_complete.py [n_conf] [n_atoms] [reps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
a = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
fc.init(0)
X, atoms, _ = syn.synthetic_ensemble(n, a, seed=2)
out = {"n": n, "a": a}
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_rmsd_and_max_all(2)
    k, t, st = ens.bench_rmsd_and_max_all(reps)
    pairs = n * (n - 1) // 2
    out.update(kernel_ms=k, total_ms_per_pass=t / reps, fixup_pairs=int(st[1]), alignments_per_s=pairs / (t / reps * 1e-3),
               frac_fp64_peak=pairs * (53 * a + 600) / (k * 1e-3) / 78.6e12)
    if n <= 4000:
        from oracle import cpu_ref as o
        R, D, _ = ens.rmsd_and_max_all()
        rng = np.random.default_rng(0)
        iu = rng.integers(0, n, 3000)
        ju = rng.integers(0, n, 3000)
        keep = iu != ju
        iu, ju = iu[keep], ju[keep]
        r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
        out.update(max_rmsd_err=float(np.abs(R[iu, ju] - r0).max()), max_dev_err=float(np.abs(D[iu, ju] - d0).max()),
                   symmetric=bool(np.array_equal(R, R.T)), diag_zero=bool(np.all(np.diag(R) == 0)))
print(json.dumps(out))
