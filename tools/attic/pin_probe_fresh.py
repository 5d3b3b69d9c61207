"""tools/attic/pin_probe_fresh.py -- the three ways to get a caller's pageable array to the device when the array is a NEW
allocation every time and is FREED right after the call (what prune_by_rmsd sees from a caller that builds its ensemble,
prunes it and drops it): (A) the runtime's own path (it pins the caller's pages and keeps them in a cache until the
memory goes away), (B) hipHostRegister + DMA + hipHostUnregister, (C) memmove into the library's pinned pieces + DMA.
Per way: the copy itself, and a 4-byte memset + wait issued right AFTER the array was freed (does the free stall the
queues?).  ctypes on libamdhip64, no library of ours."""
import ctypes as C
import gc
import json
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")


def chk(rc):
    assert rc == 0, rc


import sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000 * 50 * 3
nb = n * 8
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), nb))
pin = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(pin), nb, 0))
st = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(st), 1))


def fresh(seed):
    x = np.empty(n)
    x.fill(float(seed))  # pages touched
    return x


big = C.c_void_p(); chk(hip.hipMalloc(C.byref(big), 1 << 30))
st2 = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(st2), 1))


def tiny():
    t0 = time.perf_counter()
    chk(hip.hipMemsetAsync(d, 0, 4, st)); chk(hip.hipStreamSynchronize(st))
    return time.perf_counter() - t0


def busy_begin():  # ~1 ms of device work on another stream, in flight while the array is freed
    for _ in range(4):
        chk(hip.hipMemsetAsync(big, 1, 1 << 30, st2))
    return time.perf_counter()


def busy_end(t0):
    chk(hip.hipStreamSynchronize(st2))
    return time.perf_counter() - t0


def way(kind, reps=24):
    cp, after, busy = [], [], []
    for r in range(reps):
        x = fresh(r)
        xp = C.c_void_p(x.ctypes.data)
        t0 = time.perf_counter()
        if kind == "A":
            chk(hip.hipMemcpyAsync(d, xp, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
        elif kind == "B":
            chk(hip.hipHostRegister(xp, nb, 0)); chk(hip.hipMemcpyAsync(d, xp, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
            chk(hip.hipHostUnregister(xp))
        else:
            C.memmove(pin, xp, nb); chk(hip.hipMemcpyAsync(d, pin, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
        cp.append(time.perf_counter() - t0)
        tb = busy_begin()
        del x, xp
        gc.collect()
        busy.append(busy_end(tb))
        after.append(tiny())
    f = lambda v: [round(min(v) * 1e3, 4), round(sorted(v)[len(v) // 2] * 1e3, 4), round(max(v) * 1e3, 4)]
    return {"copy_ms_min_median_max": f(cp[2:]), "four_1GB_memsets_in_flight_during_the_free_ms_min_median_max": f(busy[2:]),
            "tiny_op_after_free_ms_min_median_max": f(after[2:])}


tiny()
print(json.dumps({"bytes": nb, "A_runtime_path": way("A"), "B_register_dma_unregister": way("B"), "C_pinned_detour": way("C"),
                  "A_again": way("A")}))
