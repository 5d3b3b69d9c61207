"""tools/cf_stamps.py -- tuning build only (make BUILD=build_stamps OUT=../libfc_hip_stamps.so EXTRA="-DFC_TUNING_BUILD -DFC_TFD_STAMPS",
FC_LIB_PATH=firecode_amd/libfc_hip_stamps.so): ticks (wall_clock64, 100 MHz) per phase of chunk_front on the cfg3 first-match
array, one chunk in 64 sampled, by chunk-length class (<= 19, 77, 307, 1 229, 4 915 structures)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
lib = L.load()
fm = np.load("tools/cfg3_fm.npz")["fm"].astype(np.int64)
N = len(fm)
out = (C.c_ulonglong * 48)()
names = ["valid + ranks", "tuple set", "roots", "sizes + members", "tiny components", "exports"]
for rep in range(2):
    mask = np.zeros(N, dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    assert lib.fc_debug_cf_stamps(out, 1) == 0
    v = [int(x) for x in out]
    for e, cls in enumerate(("<=19", "<=77", "<=307", "<=1229", "<=4915")):
        n = max(v[e * 8 + 7], 1)
        print(cls, "sampled chunks", v[e * 8 + 7], {nm: round(v[e * 8 + i] / n / 100, 2) for i, nm in enumerate(names)}, "us per chunk")
