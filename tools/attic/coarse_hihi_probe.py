"""tools/attic/coarse_hihi_probe.py -- would a hi x hi^T-only first pass of the split-half screen (18 instead of 54
MFMAs per 16 x 16 sub-tile, bounds widened by the 2^-10 s mass of the cross terms it leaves out) rule out whole
sub-tiles of the bench ensembles?  CPU estimate in float64: the polynomial values P0 / s^4 and u / s^2 of sampled
sub-tiles against the widened bounds of kabsch_f32_bounds with db = 2^-10 (1 + 2^-10) + the tight part."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from firecode_amd import synthetic as syn  # noqa: E402


def probe(X, thr=0.5, tiles=400, seed=0):
    n, A, _ = X.shape
    X = X - X.mean(axis=1, keepdims=True)
    G = (X * X).sum(axis=(1, 2))
    u24 = 2.0 ** -24
    KS2 = (A + 31) // 32
    db = 2.0 ** -10 * (1 + 2.0 ** -10) + (13.02 + 66.4 * KS2) * u24 + u24
    p0, p2 = 2 * (45 * db + 172 * u24), 2 * (6 * db + 52 * u24)
    rng = np.random.default_rng(seed)
    ok_tiles, ok_pairs, n_pairs = 0, 0, 0
    for _ in range(tiles):
        i0 = int(rng.integers(0, n // 16 - 1)) * 16
        j0 = int(rng.integers(i0 // 16 + 1, n // 16)) * 16
        P, Q = X[i0:i0 + 16], X[j0:j0 + 16]
        B = np.einsum("iax,jay->ijxy", P, Q)
        s = 0.5 * (G[i0:i0 + 16, None] + G[None, j0:j0 + 16])
        L = s - 0.5 * A * (thr * thr + 1e-6)
        n2 = (B * B).sum(axis=(2, 3))
        uu = L * L - n2
        cof = np.linalg.det(B)[..., None, None] * np.linalg.inv(B).transpose(0, 1, 3, 2)
        e2 = (cof * cof).sum(axis=(2, 3))
        P0 = uu * uu - 4 * (e2 + 2 * L * np.linalg.det(B))
        ruled_out = (uu > p2 * s * s) & (P0 > p0 * s ** 4)
        ok_tiles += bool(ruled_out.all())
        ok_pairs += int(ruled_out.sum())
        n_pairs += ruled_out.size
    return {"sub_tiles_ruled_out_whole": ok_tiles / tiles, "pairs_ruled_out": ok_pairs / n_pairs, "p0": p0, "p2": p2}


out = {}
X, _, _ = syn.synthetic_ensemble(10000, 50, seed=2)
out["cfg2_clustered_10000x50"] = probe(X)
X, _, _ = syn.synthetic_ensemble(8000, 80, seed=6)
out["cfg4_shape_8000x80"] = probe(X)
out["continuous_10000x50"] = probe(syn.continuous_ensemble(10000, 50, seed=11, thr=0.5))
print(json.dumps(out))
