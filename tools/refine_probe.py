"""Cost of candidates the screen cannot decide: an uncentred ensemble far from the origin makes the
single-precision screen pass (nearly) every pair to the exact refine (FC_SCREEN_F32=2 forces it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn

fc.init(0)
out = {}
for n in (1000, 3000):
    X, atoms, _ = syn.synthetic_ensemble(n, 50, seed=7)
    for label, Y, center in (("centred", X, True), ("offset_30A_uncentred", X - X.mean(axis=1, keepdims=True) + 30.0, False)):
        with fc.DeviceEnsemble(Y, center=center) as ens:
            ens.prune(0.5, 1.0)
            t0 = time.perf_counter()
            mask, st = ens.prune(0.5, 1.0)
            dt = time.perf_counter() - t0
        out[f"{n}_{label}"] = {"ms": dt * 1e3, "pairs": int(st[0]), "refined": int(st[1]), "similar": int(st[2]),
                               "screen": fc._lib.screen_last_kind()}
print(json.dumps({"env": os.environ.get("FC_SCREEN_F32"), **out}))
