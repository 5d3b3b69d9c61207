"""The exact refine alone (fc_bench_refine) on the continuous-RMSD ensemble of config.secondary, and the overlapped step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc
from firecode_amd import synthetic as syn
fc.init(0)
X = syn.continuous_ensemble(10000, 50, seed=11, thr=0.5)
with fc.DeviceEnsemble(X, center=True) as ens:
    ens.bench_refine(0.5, 1.0, reps=3)
    ms, n = ens.bench_refine(0.5, 1.0, reps=20)
    ens.bench_prune(0.5, 1.0, reps=2, want_mask=False)
    k, s, mask, st = ens.bench_prune(0.5, 1.0, reps=40, want_mask=True)
print(json.dumps({"refine_ms": ms, "candidates": int(n), "step_ms": s, "screen_ms_in_step": k, "similar": int(st[2]), "survivors": int(mask.sum())}))
