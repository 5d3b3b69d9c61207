"""The ladder of a long similar-pair list alone: one prune of the continuous-RMSD ensemble after another (no pipeline), kernel
times from rocprofv3 are what to read; prints the synchronous prune time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X = syn.continuous_ensemble(10000, 50, seed=11, thr=0.5)
with fc.DeviceEnsemble(X, center=True) as ens:
    for _ in range(3):
        m, st = ens.prune(0.5, 1.0)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); m, st = ens.prune(0.5, 1.0); ts.append(time.perf_counter() - t0)
print(json.dumps({"prune_ms_min": min(ts) * 1e3, "survivors": int(m.sum()), "similar": int(st[2])}))
