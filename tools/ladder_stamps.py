"""tools/ladder_stamps.py -- tuning build only (make BUILD=build_stamps OUT=../libfc_hip_stamps.so EXTRA="-DFC_TUNING_BUILD -DFC_TFD_STAMPS",
FC_LIB_PATH=firecode_amd/libfc_hip_stamps.so): cycles (s_memtime, 100 MHz ticks on gfx950) per phase of comp_group_first for the
largest component of the cfg3 ladder"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
lib = L.load()
z = np.load("tools/cfg3_fm.npz")
fm = z["fm"].astype(np.int64); N = len(fm)
out = (C.c_ulonglong * 16)()
for rep in range(3):
    mask = np.zeros(N, dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), N, L.pb(mask))
    assert lib.fc_debug_tfd_stamps(out, 1) == 0
    print([int(v) for v in out][:12])
