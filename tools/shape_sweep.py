"""tools/shape_sweep.py [n] -- the pipelined prune of clustered synthetic ensembles over atom counts and skeleton seeds:
which screen ran (16 = split-half f16, 32 = fp32 matrix pipe, 64 = fp64), ms per step, pair decisions per second.
A table to look for cliffs in the selection heuristics (round 4 found one at 80 atoms: DESIGN 5.2, the band rule)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
fc.init(0)
compact = len(sys.argv) > 3 and sys.argv[3] == "compact"  # globules (radius of gyration of folded molecules) instead of walks
atom_counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (8, 16, 24, 32, 33, 50, 64, 65, 80, 96, 97, 110, 128, 129, 160, 192, 193, 200, 224, 260, 320, 384, 416)
for a in atom_counts:
    for seed in (1, 2, 4):
        # (large structures: few cluster centres -- the generator redraws a centre until none of its atoms clash)
        X, atoms, asg = syn.synthetic_ensemble(n, a, seed=seed, cluster_size=5 if (a <= 200 or compact) else 50, compact=compact)
        G = ((X - X.mean(axis=1, keepdims=True)) ** 2).sum(axis=(1, 2)).max()
        with fc.DeviceEnsemble(X, center=True) as ens:
            ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)
            tk, ts, mask, st = ens.bench_prune(0.5, 1.0, reps=6, want_mask=True)
            kind = fc._lib.screen_last_kind()
        pairs = n * (n - 1) // 2
        print(json.dumps({"atoms": a, "seed": seed, "compact": compact, "rg": round(float((G / a) ** 0.5), 2), "screen": kind, "step_ms": round(ts, 3),
                          "screen_ms": round(tk, 3), "pair_decisions_per_s": round(pairs / ts * 1e3, -8), "candidates": int(st[1]),
                          "survivors": int(mask.sum()), "clusters": int(len(np.unique(asg)))}), flush=True)
