"""The split-half screen's assumptions and margins on this device: the f16 matrix-pipe model check
(fc_debug_mfma_f16_model), the screen's covariance accumulators against fp64 (fc_debug_h2_covariance), and
A/B timings of the three matrix-pipe screens on the cfg2 ensemble (fc_screen_select)."""
import ctypes as C
import json
import sys

sys.path.insert(0, "/root/repo")
import numpy as np

import firecode_amd as fc
from firecode_amd import _lib
from firecode_amd import synthetic as syn

fc.init(0)
flags = np.zeros(8, dtype=np.int64)
worst = C.c_double(-1.0)
_lib.call("fc_debug_mfma_f16_model", 200000, _lib.pi(flags), C.byref(worst))
print(json.dumps({"model_flags": flags.tolist(), "worst_error_in_u_of_C_plus_sum_abs_products": worst.value,
                  "charged_per_instruction": 36.0, "trials": 200000}))
for n, a in ((512, 50), (256, 13), (256, 90), (128, 128)):
    X, _, _ = syn.synthetic_ensemble(n, a, seed=7)
    Xc = X - X.mean(axis=1, keepdims=True)
    G = (Xc ** 2).sum(axis=(1, 2))
    ratios = []
    with fc.DeviceEnsemble(X, center=True) as ens:
        for ib in range(0, n, 64):
            for jb in range(0, n, 48):
                B = np.zeros(256 * 9, dtype=np.float32)
                scale, bound = C.c_double(0.0), C.c_double(0.0)
                _lib.call("fc_debug_h2_covariance", ens.handle, ib, jb, B.ctypes.data_as(C.POINTER(C.c_float)), C.byref(scale), C.byref(bound))
                ref = np.einsum("iax,jay->ijxy", Xc[ib:ib + 16], Xc[jb:jb + 16])
                s = 0.5 * (G[ib:ib + 16, None] + G[None, jb:jb + 16])
                err = np.abs(B.reshape(16, 16, 3, 3).astype(np.float64) / scale.value ** 2 - ref).max(axis=(2, 3)) / s
                ratios.append(float(err.max() / bound.value))
    print(json.dumps({"conformers": n, "atoms": a, "entry_bound_in_u": bound.value * 2 ** 24, "scale": scale.value,
                      "worst_entry_error_over_bound": max(ratios), "tiles": len(ratios)}))
X, atoms, asg = syn.synthetic_ensemble(10000, 50, seed=2)
with fc.DeviceEnsemble(X, center=True) as ens:
    for kind in (16, 32, 64, 16):
        _lib.screen_select(kind)
        ens.bench_prune(0.5, 1.0, reps=30)
        tk, ts, mask, st = ens.bench_prune(0.5, 1.0, reps=200)
        print(json.dumps({"screen": _lib.screen_last_kind(), "kernel_ms": tk, "step_ms": ts, "candidates": int(st[1]),
                          "similar": int(st[2]), "survivors": int(mask.sum())}))
    _lib.screen_select(0)
