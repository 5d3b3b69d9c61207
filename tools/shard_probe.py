"""Screen-kernel time of single ranks of a weak-scaled sharded prune, run on ONE GPU: what each
rank of an N-GPU bench would spend in its own row blocks (no exchange)."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

fc.init(0)
for world in (1, 2, 4, 8):
    n = int(round(10000 * np.sqrt(world)))
    X, atoms, asg = syn.synthetic_ensemble(n, 50, seed=2)
    with fc.DeviceEnsemble(X, center=True) as ens:
        res = {}
        for rank in sorted({0, world // 2, world - 1}):
            ens.prune_begin(0.5, 1.0, rank, world)
            ts = [ens.prune_begin(0.5, 1.0, rank, world) for _ in range(5)]
            best = min(ts, key=lambda s: s[4])
            res[rank] = {"screen_ms": best[4] * 1e-6, "owned_pairs": int(best[0]), "candidates": int(best[1])}
        print(json.dumps({"world": world, "n": n, "ranks": res}))
