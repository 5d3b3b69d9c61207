"""The screens on an ensemble WITHOUT cluster structure: 10 000 conformers of a 50-atom chain displaced
along 6 random collective modes with Gaussian amplitudes -- pair RMSDs spread continuously from 0 to ~2.5 A,
~1.5 % of the pairs below 0.5 A and a smooth density across the threshold (the synthetic bench ensemble has
none there).  Prints, per screen choice, the time of a resident prune and what the refine was given."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import firecode_amd as fc
from firecode_amd import synthetic as syn

fc.init(0)
rng = np.random.default_rng(11)
N, A, M = int(os.environ.get("BROAD_N", 10000)), int(os.environ.get("BROAD_A", 50)), 6
THR = float(os.environ.get("BROAD_THR", 0.5))  # the amplitudes scale with it: same distribution of rmsd / THR
base, _, _ = syn.synthetic_ensemble(1, A, seed=2)
modes = np.linalg.qr(rng.normal(size=(A * 3, M)))[0].T.reshape(M, A, 3)  # orthonormal displacement fields
amp = rng.normal(scale=0.35 * np.sqrt(A) * (THR / 0.5), size=(N, M))
X = base[0][None] + np.einsum("nm,mac->nac", amp, modes)
out = {"workload": "%d x %d, continuous RMSD distribution (6 collective modes), prune at %g A" % (N, A, THR)}
with fc.DeviceEnsemble(X, center=True) as ens:
    for _ in range(2):
        mask, st = ens.prune(THR, 2 * THR)
    t0 = time.perf_counter()
    for _ in range(5):
        mask, st = ens.prune(THR, 2 * THR)
    out.update({"ms_per_prune": (time.perf_counter() - t0) / 5 * 1e3, "pairs": int(st[0]), "refined": int(st[1]),
                "similar": int(st[2]), "survivors": int(mask.sum()), "screen_launched_first": fc._lib.screen_last_kind(),
                "FC_SCREEN_F32": os.environ.get("FC_SCREEN_F32")})
print(json.dumps(out))
