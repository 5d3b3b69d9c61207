"""tools/attic/hostin_fresh_probe.py -- prune_by_rmsd on a NEW copy of the configs[1] ensemble every call, the copy dropped right
after the call, and a resident prune of a small ensemble behind each drop: does freeing an uploaded array stall the
queues (round 3 saw 10-25 ms with the runtime's own upload path in the cfg3 search)?  min / median / max over 40 calls.
(It does not show here: glibc stops returning a repeatedly allocated 12 MB block to the system, so the drop is no munmap.
The cfg3 search, whose arrays differ in size from call to call, shows it: DESIGN 6.)"""
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

fc.init(0)
X0, atoms, _ = syn.synthetic_ensemble(10000, 50, seed=2)
small = fc.DeviceEnsemble(X0[:2000], center=True)
small.prune(0.5, 1.0)
fc.pruner.prune_by_rmsd(X0[:2000], atoms, 0.5)
call, after = [], []
for r in range(42):
    X = np.array(X0) + 1e-9 * r   # a new allocation, pages touched
    t0 = time.perf_counter()
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    call.append(time.perf_counter() - t0)
    del X
    gc.collect()
    t0 = time.perf_counter()
    small.prune(0.5, 1.0)
    after.append(time.perf_counter() - t0)
f = lambda v: [round(min(v) * 1e3, 3), round(sorted(v)[len(v) // 2] * 1e3, 3), round(max(v) * 1e3, 3)]
print(json.dumps({"upload_path": "runtime" if os.environ.get("FC_STAGED_UPLOADS") == "0" else "pinned pieces",
                  "prune_by_rmsd_fresh_array_ms_min_median_max": f(call[2:]),
                  "resident_prune_after_the_free_ms_min_median_max": f(after[2:]), "survivors": int(mask.sum())}))
