"""tools/attic/secondary_probe.py -- the bench's `config.secondary` workload on its own (10 000 x 50, continuous RMSD
distribution, ~1.5 % of the pairs below 0.5 A): REPS stream-ordered prunes, for rocprofv3 runs of its kernels
(k_simbits_refine on ~9e5 candidates per prune, the split-half screen at ~2 % candidates).
Usage: python tools/attic/secondary_probe.py [reps] [n_conf] [n_atoms]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
a = int(sys.argv[3]) if len(sys.argv) > 3 else 50
fc.init(0)
X = syn.continuous_ensemble(n, a, seed=11, thr=0.5)
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_prune(0.5, 1.0, reps=2, want_mask=False)
    k, s, mask, st = ens.bench_prune(0.5, 1.0, reps=reps, want_mask=True)
print(json.dumps({"n": n, "a": a, "reps": reps, "screen_kernel_ms": k, "ms_per_step": s, "candidates": int(st[1]),
                  "similar": int(st[2]), "survivors": int(mask.sum()), "screen": fc._lib.screen_last_kind()}))
