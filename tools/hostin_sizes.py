"""prune_by_rmsd(host arrays) over the ensemble size at 50 atoms: wall time per call (min / median of 30), ms."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
for n in (100, 300, 1000, 3000, 10000):
    X, atoms, _ = syn.synthetic_ensemble(n, 50, seed=2)
    for _ in range(3):
        fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    t = []
    for _ in range(30):
        t0 = time.perf_counter(); fc.pruner.prune_by_rmsd(X, atoms, 0.5); t.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"n": n, "min_ms": round(min(t), 4), "median_ms": round(float(np.median(t)), 4)}), flush=True)
