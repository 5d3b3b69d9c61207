"""tools/fm_stamps.py -- tuning build only (make BUILD=build_stamps OUT=../libfc_hip_stamps.so EXTRA="-DFC_TUNING_BUILD -DFC_TFD_STAMPS",
FC_LIB_PATH=firecode_amd/libfc_hip_stamps.so): ticks (s_memtime, 100 MHz) per phase of the first-match walk on cfg3's fingerprints"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib as L
fc.init(0)
lib = L.load()
tf = np.ascontiguousarray(np.load("tools/cfg3_tf.npz")["tf"])
N, Q = tf.shape
fm = np.zeros(N, dtype=np.int64)
out = (C.c_ulonglong * 16)()
names = ["wgs", "prologue", "(1) boxes", "(2) rows", "(3) windows", "steps", "candidates", "windows walked", "max wg ticks", "sum wg ticks", "row tests"]
for rep in range(3):
    L.call("fc_tfd_first_match", L.pf(tf), N, Q, 10.0, L.pi(fm))
    assert lib.fc_debug_fm_stamps(out, 1) == 0
    v = [int(x) for x in out]
    print({n: v[i] for i, n in enumerate(names)})
    w = max(v[0], 1)
    print("wgs", v[0], "kernel span in ticks", v[12] - v[11] if v[11] < 2**63 else None)
    print("per wg (us): prologue %.1f  boxes %.1f  rows %.1f  windows %.1f  total %.1f  max %.1f | steps %.2f cands %.1f walked %.2f rowtests %.1f"
          % (v[1] / w / 100, v[2] / w / 100, v[3] / w / 100, v[4] / w / 100, v[9] / w / 100, v[8] / 100, v[5] / w, v[6] / w, v[7] / w, v[10] / w))
