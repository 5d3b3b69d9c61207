"""How fast can 12 MB of pageable NumPy memory reach the device?  (a) hipMemcpy from pageable memory (the runtime's own
path), (b) hipHostRegister + hipMemcpyAsync + hipHostUnregister, (c) memcpy into a pinned buffer + hipMemcpyAsync,
(d) the same from pinned memory directly (the DMA alone)."""
import ctypes as C, time, sys, json
import numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(rc):
    assert rc == 0, rc
n = 10000 * 50 * 3
x = np.random.default_rng(0).normal(size=n)
nb = x.nbytes
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), nb))
pin = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(pin), nb, 0))
st = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(st), 1))
xp = C.c_void_p(x.ctypes.data)
def t(fn, reps=15):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(min(ts) * 1e3, 4), round(sorted(ts)[len(ts) // 2] * 1e3, 4)
def a():
    chk(hip.hipMemcpyAsync(d, xp, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
def b():
    chk(hip.hipHostRegister(xp, nb, 0)); chk(hip.hipMemcpyAsync(d, xp, nb, 1, st)); chk(hip.hipStreamSynchronize(st)); chk(hip.hipHostUnregister(xp))
def c():
    C.memmove(pin, xp, nb); chk(hip.hipMemcpyAsync(d, pin, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
def dd():
    chk(hip.hipMemcpyAsync(d, pin, nb, 1, st)); chk(hip.hipStreamSynchronize(st))
def mm():
    C.memmove(pin, xp, nb)
def reg():
    chk(hip.hipHostRegister(xp, nb, 0)); chk(hip.hipHostUnregister(xp))
def pieces(k, events=False):
    def f():
        step = nb // k
        for q in range(k):
            chk(hip.hipMemcpyAsync(C.c_void_p(d.value + q * step), C.c_void_p(pin.value + q * step), step, 1, st))
            if events:
                chk(hip.hipEventRecord(evs[q % 4], st))
        chk(hip.hipStreamSynchronize(st))
    return f
evs = []
for _ in range(4):
    e = C.c_void_p(); chk(hip.hipEventCreateWithFlags(C.byref(e), 2)); evs.append(e)
extra = {"pinned_in_6_pieces_ms": t(pieces(6)), "pinned_in_6_pieces_with_events_ms": t(pieces(6, True)),
         "pinned_in_12_pieces_ms": t(pieces(12)), "pinned_in_24_pieces_ms": t(pieces(24))}
print(json.dumps(extra))
print(json.dumps({"bytes": nb, "pageable_memcpy_ms": t(a), "register_dma_unregister_ms": t(b), "memmove_then_dma_ms": t(c),
                  "dma_from_pinned_ms": t(dd), "memmove_only_ms": t(mm), "register_unregister_only_ms": t(reg)}))
