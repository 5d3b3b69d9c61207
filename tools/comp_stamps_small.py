"""tools/comp_stamps_small.py -- tuning build only (make BUILD=build_stamps2 OUT=../libfc_hip_stamps2.so EXTRA="-DFC_TUNING_BUILD
-DFC_TFD_STAMPS -DFC_TFD_STAMPS_SMALL", FC_LIB_PATH=firecode_amd/libfc_hip_stamps2.so): mean microseconds per phase of
comp_group_first for the components of 19 ... 76 nodes of the cfg3 ladder (one in 64 sampled)"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
lib = L.load()
fm = np.load("tools/cfg3_fm.npz")["fm"].astype(np.int64)
N = len(fm)
out = (C.c_ulonglong * 16)()
names = ["source + shortcuts", "parent look-up", "offsets", "neighbour lists", "walk", "first set", "second set", "walk set-up"]
for rep in range(2):
    mask = np.zeros(N, dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    assert lib.fc_debug_tfd_stamps(out, 1) == 0
    v = [int(x) for x in out]
    n = max(v[11], 1)
    print("sampled", v[11], {nm: round(v[i] / n / 100, 2) for i, nm in enumerate(names)}, "us per component; sum", round(sum(v[:8]) / n / 100, 2))
