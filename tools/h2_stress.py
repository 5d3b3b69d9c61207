"""Randomised comparison of the split-half screen (kind 16) with the fp64 screen (kind 64): ensembles of random
size, atom count, threshold, noise structure and scale; similarity bits, grey counts and masks must be equal.
One-off validation run (the parity tests pin fixed cases): python tools/h2_stress.py [n_cases] [seed]"""
import json
import sys

sys.path.insert(0, "/root/repo")
import numpy as np

import firecode_amd as fc
from firecode_amd import _lib
from firecode_amd._lib import unpack_bits

fc.init(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = []
stats = {"cases": 0, "similar_pairs": 0, "declined": 0}
for case in range(n_cases):
    n = int(rng.integers(20, 1400))
    a = int(rng.choice([3, 4, 7, 13, 20, 31, 32, 33, 50, 64, 65, 80, 96, 97, 128]))
    thr = float(rng.choice([0.05, 0.2, 0.5, 1.0, 2.5]))
    kind = rng.choice(["clusters", "continuous", "line", "planar", "offset", "scaled"])
    k = max(1, n // int(rng.integers(2, 12)))
    base = rng.normal(scale=rng.uniform(0.5, 6.0), size=(k, a, 3))
    if kind == "line":
        base[:, :, 1:] *= 1e-3
    if kind == "planar":
        base[:, :, 2] *= 1e-4
    X = base[rng.integers(0, k, n)] + rng.normal(scale=thr * rng.choice([0.05, 0.3, 0.6, 1.0]), size=(n, a, 3))
    if kind == "continuous":
        X = base[0][None] + rng.normal(scale=thr * 0.7, size=(n, a, 3))
    q, r = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    q[np.linalg.det(q) < 0, :, 0] *= -1
    X = np.einsum("nij,naj->nai", q, X)
    center = True
    if kind == "offset":
        X = X + rng.normal(scale=30.0, size=3)
        center = bool(rng.integers(0, 2))
    if kind == "scaled":
        f = float(10.0 ** rng.integers(-4, 4))
        X, thr = X * f, thr * f
    out = {}
    try:
        for sel in (64, 16):
            _lib.screen_select(sel)
            try:
                with fc.DeviceEnsemble(X, center=center) as ens:
                    bits, grey = ens.simbits(thr, 2 * thr)
                    mask, st = ens.prune(thr, 2 * thr)
                out[sel] = (unpack_bits(bits, n), grey, mask)
            except fc.FirecodeHipInputError:
                assert sel == 16
                stats["declined"] += 1
    finally:
        _lib.screen_select(0)
    stats["cases"] += 1
    if 16 in out:
        stats["similar_pairs"] += int(out[64][0].sum())
        ok = np.array_equal(out[64][0], out[16][0]) and out[64][1] == out[16][1] and np.array_equal(out[64][2], out[16][2])
        if not ok:
            bad.append({"case": case, "n": n, "a": a, "thr": thr, "kind": str(kind), "center": center,
                        "bits_differ": int((out[64][0] != out[16][0]).sum())})
print(json.dumps({**stats, "mismatches": bad}))
sys.exit(1 if bad else 0)
