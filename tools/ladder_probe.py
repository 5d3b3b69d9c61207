"""tools/ladder_probe.py [reps] -- the device ladder alone on the cfg3 first-match array kept in tools/cfg3_fm.npz
(tools/dump_cfg3_fm.py): one warm-up call, then `reps` timed calls.  For rocprofv3 --kernel-trace --stats."""
import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
z = np.load("tools/cfg3_fm.npz")
fm = z["fm"].astype(np.int64)
ref = np.unpackbits(z["mask"])[:len(fm)]
N = len(fm)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ts = []
for rep in range(reps + 1):
    mask = np.zeros(N, dtype=np.uint8)
    t0 = time.perf_counter()
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    ts.append(time.perf_counter() - t0)
    assert np.array_equal(mask, ref)
print(json.dumps({"N": N, "kept": int(mask.sum()), "first_call_s": ts[0], "calls_s": ts[1:]}))
