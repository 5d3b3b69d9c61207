import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X, atoms, _ = syn.synthetic_ensemble(10000, 50, seed=2)
for _ in range(3): fc.pruner.prune_by_rmsd(X, atoms, 0.5)
ts=[]
for _ in range(20):
    t0=time.perf_counter(); fc.pruner.prune_by_rmsd(X, atoms, 0.5); ts.append(time.perf_counter()-t0)
print("min %.3f ms median %.3f ms"%(min(ts)*1e3, sorted(ts)[10]*1e3))
