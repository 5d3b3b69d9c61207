"""Where the host-in -> mask-out time of prune_by_rmsd(10 000 x 50) goes: ensemble creation (H2D + prep),
the resident prune, the output copy."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn

fc.init(0)
X, atoms, _ = syn.synthetic_ensemble(10000, 50, seed=2)
fc.pruner.prune_by_rmsd(X, atoms, 0.5)
out = {"create_ms": [], "prune_ms": [], "index_ms": [], "total_ms": []}
for _ in range(7):
    t0 = time.perf_counter()
    ens = fc.DeviceEnsemble(X, center=True)
    t1 = time.perf_counter()
    mask, st = ens.prune(0.5, 1.0)
    t2 = time.perf_counter()
    kept = X[mask]
    t3 = time.perf_counter()
    ens.close()
    out["create_ms"].append((t1 - t0) * 1e3); out["prune_ms"].append((t2 - t1) * 1e3); out["index_ms"].append((t3 - t2) * 1e3)
    t0 = time.perf_counter(); fc.pruner.prune_by_rmsd(X, atoms, 0.5); out["total_ms"].append((time.perf_counter() - t0) * 1e3)
print(json.dumps({k: round(min(v), 4) for k, v in out.items()}))
