"""tools/attic/prune_probe.py -- REPS stream-ordered prunes of a clustered synthetic ensemble (the bench's prune_path
workload or a cfg4-family member), for rocprofv3 runs.  Usage: python tools/attic/prune_probe.py [n_conf] [n_atoms] [seed] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
a = int(sys.argv[2]) if len(sys.argv) > 2 else 50
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
fc.init(0)
X, atoms, assign = syn.synthetic_ensemble(n, a, seed=seed)
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_prune(0.5, 1.0, reps=2, want_mask=False)
    k, s, mask, st = ens.bench_prune(0.5, 1.0, reps=reps, want_mask=True)
print(json.dumps({"n": n, "a": a, "reps": reps, "screen_kernel_ms": k, "ms_per_step": s, "candidates": int(st[1]),
                  "similar": int(st[2]), "survivors": int(mask.sum()), "screen": fc._lib.screen_last_kind()}))
