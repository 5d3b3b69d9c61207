"""tools/time_complete.py -- time the complete all-pairs alignment kernel (fc_bench_rmsd_and_max_all) on the
BASELINE configs[1] ensemble (values: tests/test_gpu_fullsize.py, bench.py's value_check).
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
# (the generator redraws a cluster centre until no two of its atoms are closer than 0.5 A: hopeless above ~300 atoms with
# thousands of centres -- large structures get a handful of centres, which changes nothing for a kernel that does the
# same work for every pair)
X, atoms, _ = syn.synthetic_ensemble(n, a, seed=2, cluster_size=5 if a <= 200 else max(5, n // 8))
out = {"n": n, "a": a}
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_rmsd_and_max_all(2)
    k, t, st = ens.bench_rmsd_and_max_all(reps)
    pairs = n * (n - 1) // 2
    out.update(kernel_ms=k, total_ms_per_pass=t / reps, fixup_pairs=int(st[1]), alignments_per_s=pairs / (t / reps * 1e-3),
               frac_fp64_peak=pairs * (53 * a + 600) / (k * 1e-3) / 78.6e12)
print(json.dumps(out))
