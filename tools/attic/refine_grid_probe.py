"""tools/attic/refine_grid_probe.py -- k_refine_pairs alone (fc_bench_refine) on the continuous-RMSD ensemble."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X = syn.continuous_ensemble(10000, 50, seed=11, thr=0.5)
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_refine(0.5, 1.0, reps=2)
    ms, n = ens.bench_refine(0.5, 1.0, reps=20)
print(json.dumps({"grid_per_cu": os.environ.get("FC_REFINE_GRID", "8"), "refine_ms": ms, "candidates": n,
                  "alignments_per_s": n / (ms * 1e-3), "hbm_frac_8d": n * 2416 / (ms * 1e-3) / 8e12}))
