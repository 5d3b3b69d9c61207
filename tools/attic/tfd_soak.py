"""Repeats the device-built TFD ladder (three levels in flight, arena blocks, compact downloads) against the all-host ladder
on changing first-match arrays in ONE process: a race between the helper streams would show as a mask that differs."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib
from test_tfd_gpu_graph import _random_first_match

fc.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad, t0 = 0, time.perf_counter()
for it in range(reps):
    n = (200000, 524288, 1000003, 1679611)[it % 4]
    kind = ("mixed", "near", "stars", "chain")[(it // 4) % 4]
    fm = _random_first_match(np.random.default_rng(100 + it), n, kind)
    out = {}
    for env in ("1", "0"):
        os.environ["FC_TFD_GPU"] = env
        m = np.zeros(n, dtype=np.uint8)
        _lib.call("fc_tfd_ladder_from_first_match", _lib.pi(fm), n, _lib.pb(m))
        out[env] = m
    same = bool(np.array_equal(out["0"], out["1"]))
    bad += not same
    print(json.dumps({"it": it, "n": n, "kind": kind, "same": same, "kept": int(out["0"].sum())}), flush=True)
print(json.dumps({"reps": reps, "mismatches": bad, "seconds": round(time.perf_counter() - t0, 1)}))
sys.exit(1 if bad else 0)
