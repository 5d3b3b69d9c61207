import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X, atoms, _ = syn.synthetic_ensemble(10000, 50, seed=2)
for _ in range(3): fc.pruner.prune_by_rmsd(X, atoms, 0.5)
for _ in range(6): fc.pruner.prune_by_rmsd(X, atoms, 0.5)
