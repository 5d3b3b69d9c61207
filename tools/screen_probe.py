import os, sys, json
sys.path.insert(0, '/root/repo')
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X, atoms, asg = syn.synthetic_ensemble(10000, 50, seed=2)
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_prune(0.5, 1.0, reps=30)
    tk, ts, mask, st = ens.bench_prune(0.5, 1.0, reps=200)
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("FC_")}, "kernel_ms": tk, "step_ms": ts, "stats": st.tolist(), "survivors": int(mask.sum())}))
