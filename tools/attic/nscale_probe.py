"""Screen-kernel efficiency against ensemble size and atom count (tuning probe)."""
import json
import sys

sys.path.insert(0, ".")
import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402

fc.init(0)
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(5000, 50), (10000, 50), (20000, 50),
                                                                         (40000, 50), (10000, 100), (20000, 100)]
for n, a in cases:
    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=2)
    with fc.DeviceEnsemble(X, center=True) as ens:
        ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)
        ks = [ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)[0] for _ in range(7)]
    k = min(ks)
    pairs = n * (n - 1) // 2
    a4 = (a + 3) // 4 * 4
    print(json.dumps({"n": n, "a": a, "kernel_ms": k, "frac": pairs * 18 * a4 / (k * 1e-3) / 78.6e12}))
