import sys, time, json, os
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
from itertools import product
rng = np.random.default_rng(0)
for T, vals in ((11, (-120.0, 0.0, 120.0)), (9, (-150.0, -90.0, -30.0, 30.0))):
    grid = np.array(np.meshgrid(*[vals] * T, indexing="ij")).reshape(T, -1).T.copy()
    # systematic order, with back-off-like perturbations: a third of the rows lose 5 or 10 degrees on one angle, and
    # 20 % of the rows repeat an earlier row (clashes undone)
    n = len(grid)
    tf = grid + rng.choice([0.0, 0.0, -5.0, -10.0], size=grid.shape) * (rng.random(grid.shape) < 0.15)
    rep = rng.random(n) < 0.2
    src = np.maximum(np.arange(n) - rng.integers(1, 2000, n), 0)
    tf[rep] = tf[src[rep]]
    tf = np.ascontiguousarray(tf)
    out = {}
    for name, env in (("u16", {}), ("f32", {"FC_TFD_U16": "0"})):
        os.environ.pop("FC_TFD_U16", None)
        os.environ.update(env)
        fm = np.zeros(n, dtype=np.int64)
        ts = []
        for rep_ in range(4):
            t0 = time.perf_counter()
            L.call("fc_tfd_first_match", L.pf(tf), n, T, 10.0, L.pi(fm))
            ts.append(time.perf_counter() - t0)
        out[name] = fm
        print(T, n, name, [round(1e3 * t, 2) for t in ts], int((fm >= 0).sum()))
    os.environ.pop("FC_TFD_U16", None)
    print("equal", bool(np.array_equal(out["u16"], out["f32"])))
