"""ctypes binding of libfc_hip.so (include/fc_hip.h).

The library is the product: there is no Python/NumPy fallback.  If the shared
object is missing, or no gfx950 device is usable, the calls raise
``FirecodeHipError`` -- loudly, never silently.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FC_LIB_PATH") or os.path.join(_HERE, "libfc_hip.so")  # FC_LIB_PATH: a tuning build

FC_OK = 0
FC_E_INVALID, FC_E_NODEVICE, FC_E_HIP, FC_E_NOMEM, FC_E_LIMIT = -1, -2, -3, -4, -5


class FirecodeHipError(RuntimeError):
    """Any failure reported by libfc_hip.so (code in ``.code``)."""

    def __init__(self, code, message):
        super().__init__(f"libfc_hip error {code}: {message}")
        self.code = code


class FirecodeHipInputError(FirecodeHipError, ValueError):
    """FC_E_INVALID / FC_E_LIMIT: the arguments were rejected before any launch."""


class FirecodeHipDeviceError(FirecodeHipError):
    """FC_E_NODEVICE / FC_E_HIP / FC_E_NOMEM."""


_lib = None

_p_f64 = C.POINTER(C.c_double)
_p_i64 = C.POINTER(C.c_int64)
_p_u64 = C.POINTER(C.c_uint64)
_p_u8 = C.POINTER(C.c_uint8)
_i64 = C.c_int64
_f64 = C.c_double
_ens = C.c_void_p

# name -> argtypes  (restype is always int unless noted)
_SIGNATURES = {
    "fc_abi_version": [],
    "fc_device_count": [],
    "fc_init": [C.c_int],
    "fc_shutdown": [],
    "fc_warmup": [],
    "fc_device_info": [C.c_char_p, _i64, _p_i64, _p_i64],
    "fc_ensemble_create": [_p_f64, _i64, _i64, _p_u8, C.c_int, C.POINTER(_ens)],
    "fc_ensemble_destroy": [_ens],
    "fc_ensemble_shape": [_ens, _p_i64, _p_i64],
    "fc_kabsch_rmsd_pairs": [_p_f64, _i64, _i64, _p_u8, _p_i64, _p_i64, _i64, C.c_int, _p_f64, _p_f64],
    "fc_ensemble_rmsd_pairs": [_ens, _p_i64, _p_i64, _i64, _p_f64, _p_f64],
    "fc_ensemble_rmsd_matrix": [_ens, _p_f64, _p_f64],
    "fc_ensemble_rmsd_values": [_ens, _p_f64, _p_f64],
    "fc_ensemble_rmsd_and_max_all": [_ens, _p_f64, _p_f64, _p_f64],
    "fc_bench_rmsd_and_max_all": [_ens, _i64, _p_f64, _p_f64, _p_i64],
    "fc_bench_rmsd_and_max_all_sampled": [_ens, _i64, _p_i64, _p_i64, _i64, _p_f64, _p_f64, _p_f64, _p_f64, _p_i64],
    "fc_bench_refine": [_ens, _f64, _f64, _i64, _p_f64, _p_i64],
    "fc_screen_select": [C.c_int],
    "fc_prune_conventions": [C.c_int],
    "fc_prune_similarity": [_p_f64, _i64, _i64, _p_u8, _p_f64, C.c_int, _f64, C.c_int, _f64, _f64, _p_f64, _f64, _i64,
                            _p_u8, _p_u8, _p_i64],
    "fc_alignment_matrices": [_p_f64, _p_f64, _i64, _i64, _p_f64],
    "fc_rmsd_simbits": [_ens, _f64, _f64, _p_f64, _f64, _i64, _i64, _p_u64, _p_i64],
    "fc_prune_rmsd": [_ens, _f64, _f64, _p_f64, _f64, _i64, _p_u8, _p_i64],
    "fc_prune_rmsd_host": [_p_f64, _i64, _i64, _p_u8, C.c_int, _f64, _f64, _p_f64, _f64, _i64, _p_u8, _p_i64],
    "fc_greedy_prune_from_bits": [_p_u64, _i64, _i64, _p_u8],
    "fc_prune_rmsd_begin": [_ens, _f64, _f64, _p_f64, _f64, _i64, _i64, _i64, _p_i64],
    "fc_prune_level": [_ens, _i64, _p_u8, _p_u8],
    "fc_prune_similar_pairs": [_ens, _p_u64, _i64, _p_i64],
    "fc_prune_from_pairs": [_ens, _p_u64, _i64, _i64, _p_u8],
    "fc_prune_rmsd_begin_async": [_ens, _f64, _f64, _i64, _i64, _i64],
    "fc_prune_export_pairs_dev": [_ens, C.c_void_p, _i64],
    "fc_prune_from_gathered_dev": [_ens, C.c_void_p, _i64, _i64, _i64, _p_u8, _p_i64],
    "fc_prune_from_gathered_dev_enqueue": [_ens, C.c_void_p, _i64, _i64, _i64, _i64, _i64],
    "fc_prune_collect": [_ens, _i64, _i64, _p_u8, _p_i64],
    "fc_stream_set": [C.c_void_p],
    "fc_memory_trim": [],
    "fc_host_alloc_pinned": [C.c_int64, C.POINTER(C.c_void_p)],
    "fc_host_free_pinned": [C.c_void_p],
    "fc_inertia_moments": [_p_f64, _i64, _i64, _p_f64, _p_f64],
    "fc_prune_rmsd_rot_corr": [_p_f64, _i64, _i64, _p_u8, _p_i64, _i64, _p_u8, _p_f64, C.POINTER(C.c_int32), _i64,
                               _f64, _f64, _p_f64, _f64, _i64, _p_u8, _p_u64],
    "fc_align_by_moi": [_p_f64, _i64, _i64, _p_f64, _p_f64],
    "fc_prune_moi": [_p_f64, _i64, _i64, _p_f64, _f64, _p_f64, _f64, _i64, _p_u8],
    "fc_align_to_first": [_p_f64, _i64, _i64, _p_i64, _i64, _p_f64],
    "fc_rototranslate": [_p_f64, _i64, _i64, _p_f64, _p_f64, _p_f64],
    "fc_clash_self": [_p_f64, _i64, _i64, _f64, _f64, _p_i64],
    "fc_clash_fragments": [_p_f64, _i64, _i64, _p_i64, _i64, _f64, _i64, _p_i64, _p_u8],
    "fc_clash_graph": [_p_f64, _i64, _i64, _p_u8, _f64, _p_i64],
    "fc_fitness_check": [_p_f64, _i64, _i64, _p_i64, _p_f64, _i64, _f64, _p_f64, _p_u8],
    "fc_embed_poses_clash": [_p_f64, _i64, _i64, _p_f64, _i64, _i64, _p_i64, _p_i64, _p_f64, _p_f64,
                             _p_f64, _p_f64, _i64, _f64, _i64, _p_i64, _p_u8, _p_f64],
    "fc_embed_mol_transforms": [_p_f64, _i64, _i64, _p_i64, _i64, _p_f64, _p_f64, _i64, _p_f64, _i64, _p_f64, _p_f64],
    "fc_embed_trimolecular": [C.POINTER(_p_f64), _p_i64, _p_i64, C.POINTER(_p_i64), _p_i64, _i64, _p_i64, _p_f64,
                              _p_f64, _p_f64, _p_f64, _p_u8, _p_i64, _p_f64, _p_f64, _i64, C.POINTER(C.c_int32),
                              _i64, _f64, _i64, _f64, _p_f64, _p_f64, _p_u8, _p_u8],
    "fc_embed_grid_clash": [_p_f64, _i64, _i64, _p_i64, _i64, _p_f64, _p_f64, _p_f64, _i64, _i64, _p_i64, _i64,
                            _p_f64, _p_f64, _p_f64, _i64, _p_f64, _i64, _f64, _i64, _p_u8,
                            C.POINTER(C.c_int32), _p_f64],
    "fc_embed_grid_dedupe": [_p_f64, _i64, _i64, _p_i64, _i64, _p_f64, _p_f64, _p_f64, _i64, _i64, _p_i64, _i64,
                             _p_f64, _p_f64, _p_f64, _i64, _p_f64, _i64, _f64, _i64, _f64, _p_u8, _p_u8],
    "fc_string_embed": [_p_f64, _i64, _i64, _p_f64, _p_f64, _i64, _p_f64, _i64, _i64, _p_f64, _p_f64, _i64, _p_f64, _i64,
                        _p_i64, _i64, _f64, _i64, _f64, _p_u8, _p_u8, _p_f64, _p_f64],
    "fc_torsion_scan": [_p_f64, _i64, _p_i64, _i64, _p_u8, _p_i64, _i64, _f64, _i64, _p_f64, _p_i64],
    "fc_torsion_scan_fingerprints": [_p_f64, _i64, _p_i64, _i64, _p_u8, _p_i64, _i64, _f64, _i64, _p_i64, _i64,
                                     _p_f64, _p_i64, _p_f64],
    "fc_torsion_scan_tfd": [_p_f64, _i64, _p_i64, _i64, _p_u8, _p_i64, _i64, _f64, _i64, _p_i64, _i64, _f64, _p_i64, _p_u8],
    "fc_torsion_scan_tfd_grid": [_p_f64, _i64, _p_i64, _i64, _p_u8, _p_i64, _p_i64, _f64, _i64, _p_i64, _i64, _f64, _p_i64, _p_u8],
    "fc_torsion_fingerprint": [_p_f64, _i64, _i64, _p_i64, _i64, _p_f64],
    "fc_tfd_simbits": [_p_f64, _i64, _i64, _f64, _i64, _i64, _p_u64],
    "fc_tfd_first_match": [_p_f64, _i64, _i64, _f64, _p_i64],
    "fc_tfd_ladder_from_first_match": [_p_i64, _i64, _p_u8],
    "fc_tfd_prune": [_p_f64, _i64, _i64, _f64, _p_u8],
    "fc_debug_pyset_order_ints": [_p_i64, _i64, _p_i64, _p_i64],
    "fc_debug_pyset_order_pairs": [_p_i64, _i64, _p_i64, _p_i64],
    "fc_debug_pyset_order_pairs_device": [_p_i64, _i64, _p_i64],
    "fc_debug_tfd_ladder_emulate": [_p_i64, _i64, _p_u8],
    "fc_xyz_write": [C.c_char_p, C.POINTER(C.c_char_p), _i64, _p_f64, _i64, C.c_char_p, C.c_int],
    "fc_xyz_scan": [C.c_char_p, _p_i64, _p_i64],
    "fc_cartesian_product_i64": [_p_i64, _p_i64, _i64, _p_i64],
    "fc_cartesian_product_f64": [_p_f64, _p_i64, _i64, _p_f64],
    "fc_xyz_read": [C.c_char_p, _i64, _i64, C.c_char_p, _p_f64],
    "fc_bench_prune_rmsd": [_ens, _f64, _f64, _i64, _p_f64, _p_f64, _p_u8, _p_i64],
    "fc_stream_use": [C.c_void_p],
    "fc_ensemble_twin": [_ens, C.POINTER(_ens)],
    "fc_prune_rmsd_begin_split_async": [_ens, _f64, _f64, _i64, _i64, _i64, C.c_void_p, C.c_int],
    "fc_screen_last_kind": [],
    "fc_debug_mfma_f16_model": [_i64, _p_i64, _p_f64],
    "fc_debug_h2_covariance": [_ens, _i64, _i64, C.POINTER(C.c_float), _p_f64, _p_f64],
    "fc_comm_unique_id": [_p_u8],
    "fc_comm_init": [_i64, _i64, _p_u8],
    "fc_comm_destroy": [],
    "fc_comm_info": [_p_i64, _p_i64],
    "fc_allgather_mask": [_p_u8, _i64, _p_u8],
    "fc_allgather_u8_dev": [C.c_void_p, C.c_void_p, _i64],
    "fc_comm_barrier": [],
    "fc_debug_comm_loopback": [_i64, _i64],
    "fc_prune_rmsd_sharded": [_ens, _f64, _f64, _i64, _i64, _p_u8, _p_i64],
    "fc_bench_prune_rmsd_sharded": [_ens, _f64, _f64, _i64, C.c_int, _p_f64, _p_f64, _p_u8, _p_i64],
    "fc_prune_rmsd_many": [C.POINTER(_ens), _i64, _f64, _f64, _i64, C.POINTER(_p_u8), _p_i64],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES) + ("fc_last_error",)


def _hip_runtime_mode():
    """Which HIP runtime libfc_hip.so binds to.  Default ``system``: the ROCm install the library was
    built against -- the product does not depend on PyTorch being installed.  ``FC_HIP_RUNTIME=torch``
    (explicit opt-in, for callers that hand the library torch streams / tensors: the legacy
    torch.distributed exchange in ``firecode_amd.dist``) maps the ``libamdhip64.so`` bundled with
    PyTorch's ROCm wheel first, so that both sides share ONE runtime and one device context; two HIP
    runtimes in one process cannot both own the GPU.  Importing torch BEFORE this module has the same
    effect (its copy is then already mapped under the same soname) and is reported as ``torch``."""
    want = os.environ.get("FC_HIP_RUNTIME", "system")
    if want not in ("system", "torch", "auto"):
        raise FirecodeHipInputError(FC_E_INVALID, f"FC_HIP_RUNTIME={want!r}: expected 'system' or 'torch'")
    return "system" if want == "auto" else want


def _mapped_hip_runtimes():
    """paths of every libamdhip64 already mapped into this process"""
    try:
        with open("/proc/self/maps") as fh:
            return sorted({ln.split()[-1] for ln in fh if "libamdhip64" in ln})
    except OSError:
        return []


HIP_RUNTIME = None  # "system" | "torch": set by load()


def _share_torch_hip_runtime():
    """FC_HIP_RUNTIME=torch: map torch's bundled HIP runtime ahead of libfc_hip.so (see
    ``_hip_runtime_mode``).  Refuses when another libamdhip64 is already mapped (a second copy
    would leave one side without a device) or when the bundled file does not carry the soname
    libfc_hip.so asks for."""
    import importlib.util
    import sys

    mapped = _mapped_hip_runtimes()
    if "torch" in sys.modules:
        return mapped[0] if mapped else None  # torch's copy is in: libfc_hip.so binds to it by soname
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        raise FirecodeHipDeviceError(FC_E_NODEVICE, "FC_HIP_RUNTIME=torch but PyTorch is not installed")
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if mapped and os.path.realpath(mapped[0]) != os.path.realpath(path):
        raise FirecodeHipDeviceError(
            FC_E_NODEVICE, f"FC_HIP_RUNTIME=torch but {mapped[0]} is already mapped: a second HIP runtime "
            "in this process would see no GPU")
    if not os.path.exists(path):
        raise FirecodeHipDeviceError(FC_E_NODEVICE, f"FC_HIP_RUNTIME=torch but {path} does not exist")
    with open(path, "rb") as fh:  # the soname libfc_hip.so was linked against must be the one it carries
        if b"libamdhip64.so.7" not in fh.read():
            raise FirecodeHipDeviceError(
                FC_E_NODEVICE, f"{path} does not carry the soname libamdhip64.so.7 that libfc_hip.so binds to")
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


def load():
    """dlopen the library (no device is touched) and set the prototypes."""
    global _lib, HIP_RUNTIME
    if _lib is not None:
        return _lib
    import sys

    mode = _hip_runtime_mode()
    if mode == "torch":
        _share_torch_hip_runtime()
    HIP_RUNTIME = "torch" if (mode == "torch" or "torch" in sys.modules) else "system"
    if not os.path.exists(LIB_PATH):
        raise FirecodeHipDeviceError(
            FC_E_NODEVICE,
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C firecode_amd/csrc`; firecode_amd has no CPU fallback",
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.fc_last_error.argtypes = []
    lib.fc_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc == FC_OK:
        return
    msg = load().fc_last_error().decode("utf-8", "replace")
    if rc in (FC_E_INVALID, FC_E_LIMIT):
        raise FirecodeHipInputError(rc, msg)
    raise FirecodeHipDeviceError(rc, msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


# ---- array helpers ---------------------------------------------------------------
def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def ptr(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def pf(a):
    return ptr(a, C.c_double)


def pi(a):
    return ptr(a, C.c_int64)


def pb(a):
    return ptr(a, C.c_uint8)


def pw(a):
    return ptr(a, C.c_uint64)


def init(device=0):
    call("fc_init", int(device))


def shutdown():
    call("fc_shutdown")


def stream_set(hip_stream):
    """Enqueue on the caller's HIP stream (integer handle, e.g. ``torch.cuda.Stream().cuda_stream``);
    None / 0 switches back to the library's own stream."""
    call("fc_stream_set", C.c_void_p(int(hip_stream) if hip_stream else None))


def stream_use(hip_stream):
    """``stream_set`` without draining the previous stream (the caller orders streams with events)."""
    call("fc_stream_use", C.c_void_p(int(hip_stream) if hip_stream else None))


def screen_last_kind():
    """32 / 64 / 1: arithmetic of the all-pairs screen the last prune launched (fc_screen_last_kind)."""
    return int(load().fc_screen_last_kind())


def screen_select(kind=0):
    """0 = automatic, 32 = single-precision screen, 64 = fp64 screen for the prunes that follow."""
    call("fc_screen_select", int(kind))


# ---- the exchange (RCCL behind the C ABI; firecode_amd/dist.py holds the bootstrap) ------------
def comm_unique_id():
    buf = np.zeros(128, dtype=np.uint8)
    call("fc_comm_unique_id", pb(buf))
    return buf.tobytes()


def comm_init(rank, world, unique_id):
    buf = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
    if buf.shape[0] != 128:
        raise FirecodeHipInputError(FC_E_INVALID, "the RCCL unique id has 128 bytes")
    call("fc_comm_init", int(rank), int(world), pb(buf))


def comm_destroy():
    call("fc_comm_destroy")


def comm_info():
    r, w = C.c_int64(0), C.c_int64(1)
    call("fc_comm_info", C.byref(r), C.byref(w))
    return r.value, w.value


def comm_barrier():
    call("fc_comm_barrier")


def allgather_mask(mask_u8):
    """(n,) uint8 on every rank -> (world, n) uint8 (fc_allgather_mask; a group of one without comm_init)."""
    m = u8(mask_u8).reshape(-1)
    _, world = comm_info()
    out = np.zeros((world, m.shape[0]), dtype=np.uint8)
    call("fc_allgather_mask", pb(m), int(m.shape[0]), pb(out))
    return out


def warmup():
    """Load every translation unit's device code and prime the buffer pool now (fc_warmup)."""
    call("fc_warmup")


def memory_trim():
    """Return the device blocks kept by the library's caching pool to the HIP runtime."""
    call("fc_memory_trim")


def device_count():
    return int(load().fc_device_count())


def device_info():
    name = C.create_string_buffer(160)
    ncu, hbm = C.c_int64(0), C.c_int64(0)
    call("fc_device_info", name, 160, C.byref(ncu), C.byref(hbm))
    return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": hbm.value}


class DeviceEnsemble:
    """HBM-resident prepared ensemble (fc_ensemble)."""

    def __init__(self, coords, atom_mask=None, center=True):
        coords = f64(coords)
        if coords.ndim != 3 or coords.shape[2] != 3:
            raise FirecodeHipInputError(FC_E_INVALID, f"coords must be (N, A, 3), got {coords.shape}")
        self.N, self.A_all = int(coords.shape[0]), int(coords.shape[1])
        m = None if atom_mask is None else u8(np.asarray(atom_mask, dtype=bool))
        if m is not None and m.shape != (self.A_all,):
            raise FirecodeHipInputError(FC_E_INVALID, "atom_mask must have one entry per atom")
        self._h = _ens()
        call("fc_ensemble_create", pf(coords), self.N, self.A_all, pb(m), int(bool(center)), C.byref(self._h))
        self.W = (max(self.N, 1) + 63) // 64

    def close(self):
        if getattr(self, "_h", None):
            load().fc_ensemble_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        if not self._h:
            raise FirecodeHipInputError(FC_E_INVALID, "ensemble already destroyed")
        return self._h

    def rmsd_pairs(self, pair_i, pair_j):
        pi_, pj_ = i64(pair_i), i64(pair_j)
        P = int(pi_.shape[0])
        r, m = np.empty(P), np.empty(P)
        call("fc_ensemble_rmsd_pairs", self.handle, pi(pi_), pi(pj_), P, pf(r), pf(m))
        return r, m

    def rmsd_matrix(self):
        r = np.zeros((self.N, self.N))
        m = np.zeros((self.N, self.N))
        call("fc_ensemble_rmsd_matrix", self.handle, pf(r), pf(m))
        return r, m

    def rmsd_values(self, want_matrix=True):
        """All-pairs RMSD values (no max deviation) on the matrix pipe:
        (matrix (N, N) or None, kernel ms)."""
        r = np.zeros((self.N, self.N)) if want_matrix else None
        ms = C.c_double(0)
        call("fc_ensemble_rmsd_values", self.handle, pf(r), C.byref(ms))
        return r, ms.value

    def rmsd_and_max_all(self, want_matrices=True):
        """Complete alignment of all pairs: ((rmsd, maxdev) matrices or (None, None), kernel ms)."""
        r = np.zeros((self.N, self.N)) if want_matrices else None
        m = np.zeros((self.N, self.N)) if want_matrices else None
        ms = C.c_double(0)
        call("fc_ensemble_rmsd_and_max_all", self.handle, pf(r), pf(m), C.byref(ms))
        return r, m, ms.value

    def bench_refine(self, max_rmsd, max_dev, reps=10):
        """The exact refine alone over the candidate queue one screen leaves -> (mean kernel ms, candidates)."""
        ms, n = C.c_double(0), C.c_int64(0)
        call("fc_bench_refine", self.handle, float(max_rmsd), float(max_dev), int(reps), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def bench_rmsd_and_max_all(self, reps=1):
        """``reps`` complete all-pairs alignment passes, outputs resident ->
        (mean kernel ms, total ms, stats [pairs, fix-up pairs, tiled])."""
        k, t = C.c_double(0), C.c_double(0)
        stats = np.zeros(3, dtype=np.int64)
        call("fc_bench_rmsd_and_max_all", self.handle, int(reps), C.byref(k), C.byref(t), pi(stats))
        return k.value, t.value, stats

    def bench_rmsd_and_max_all_sampled(self, pair_i, pair_j, reps=1):
        """``bench_rmsd_and_max_all`` + the elements (pair_i, pair_j) of the last pass's two output matrices
        -> (mean kernel ms, total ms, stats, rmsd (P,), maxdev (P,))."""
        pi_, pj_ = i64(pair_i), i64(pair_j)
        P = int(pi_.shape[0])
        r, m = np.empty(P), np.empty(P)
        k, t = C.c_double(0), C.c_double(0)
        stats = np.zeros(3, dtype=np.int64)
        call("fc_bench_rmsd_and_max_all_sampled", self.handle, int(reps), pi(pi_), pi(pj_), P, pf(r), pf(m),
             C.byref(k), C.byref(t), pi(stats))
        return k.value, t.value, stats, r, m

    def simbits(self, max_rmsd, max_dev, energies=None, max_dE=0.0, row_begin=0, row_end=None):
        row_end = self.N if row_end is None else int(row_end)
        bits = np.zeros((row_end - row_begin, self.W), dtype=np.uint64)
        grey = C.c_int64(0)
        en = None if energies is None else f64(energies)
        call("fc_rmsd_simbits", self.handle, float(max_rmsd), float(max_dev), pf(en), float(max_dE),
             int(row_begin), row_end, pw(bits), C.byref(grey))
        return bits, grey.value

    def prune(self, max_rmsd, max_dev, energies=None, max_dE=0.0, min_per_group=20):
        mask = np.zeros(self.N, dtype=np.uint8)
        stats = np.zeros(6, dtype=np.int64)
        en = None if energies is None else f64(energies)
        call("fc_prune_rmsd", self.handle, float(max_rmsd), float(max_dev), pf(en), float(max_dE),
             int(min_per_group), pb(mask), pi(stats))
        return mask.astype(bool), stats

    def prune_begin(self, max_rmsd, max_dev, rank, world, row_block=128, energies=None, max_dE=0.0):
        stats = np.zeros(6, dtype=np.int64)
        en = None if energies is None else f64(energies)
        call("fc_prune_rmsd_begin", self.handle, float(max_rmsd), float(max_dev), pf(en), float(max_dE),
             int(rank), int(world), int(row_block), pi(stats))
        return stats

    def prune_level(self, k, mask_in):
        mi = u8(mask_in)
        mo = np.zeros(self.N, dtype=np.uint8)
        call("fc_prune_level", self.handle, int(k), pb(mi), pb(mo))
        return mo

    def similar_pairs(self, count_hint=None):
        """This rank's exactly-similar pairs after ``prune_begin`` as uint64
        ``(i << 32) | j``; raises FirecodeHipInputError (FC_E_LIMIT) when the
        candidate queue overflowed (dense similarity: use ``prune_level``).
        ``count_hint``: ``stats[2]`` of ``prune_begin`` saves the size query."""
        n = C.c_int64(0)
        if count_hint is None:
            call("fc_prune_similar_pairs", self.handle, None, 0, C.byref(n))
            count_hint = n.value
        out = np.zeros(int(count_hint), dtype=np.uint64)
        call("fc_prune_similar_pairs", self.handle, pw(out), out.shape[0], C.byref(n))
        return out[: n.value]

    def prune_from_pairs(self, pairs, min_per_group=20):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
        mask = np.zeros(self.N, dtype=np.uint8)
        call("fc_prune_from_pairs", self.handle, pw(pairs), int(pairs.shape[0]), int(min_per_group), pb(mask))
        return mask.astype(bool)

    # device-resident exchange: device pointers are plain integers (e.g. torch's data_ptr())
    def prune_begin_async(self, max_rmsd, max_dev, rank, world, row_block=128):
        call("fc_prune_rmsd_begin_async", self.handle, float(max_rmsd), float(max_dev), int(rank), int(world),
             int(row_block))

    def prune_begin_split_async(self, max_rmsd, max_dev, rank, world, screen_stream, row_block=128, timed=True):
        call("fc_prune_rmsd_begin_split_async", self.handle, float(max_rmsd), float(max_dev), int(rank),
             int(world), int(row_block), C.c_void_p(int(screen_stream)), int(bool(timed)))

    def twin(self):
        """Second prune workspace over the same resident coordinates (owned by this ensemble)."""
        t = getattr(self, "_twin", None)
        if t is None:
            h = _ens()
            call("fc_ensemble_twin", self.handle, C.byref(h))
            t = self._twin = _EnsembleView(self, h)
        return t

    def export_pairs_dev(self, dev_ptr, cap):
        call("fc_prune_export_pairs_dev", self.handle, C.c_void_p(int(dev_ptr)), int(cap))

    def prune_from_gathered_dev(self, dev_ptr, world, cap, min_per_group=20):
        mask = np.zeros(self.N, dtype=np.uint8)
        stats = np.zeros(6, dtype=np.int64)
        call("fc_prune_from_gathered_dev", self.handle, C.c_void_p(int(dev_ptr)), int(world), int(cap),
             int(min_per_group), pb(mask), pi(stats))
        return mask.astype(bool), stats

    def prune_from_gathered_enqueue(self, dev_ptr, world, cap, slot, n_slots, min_per_group=20):
        call("fc_prune_from_gathered_dev_enqueue", self.handle, C.c_void_p(int(dev_ptr)), int(world), int(cap),
             int(min_per_group), int(slot), int(n_slots))

    def prune_collect(self, slot, n_slots):
        mask = np.zeros(self.N, dtype=np.uint8)
        stats = np.zeros(6, dtype=np.int64)
        call("fc_prune_collect", self.handle, int(slot), int(n_slots), pb(mask), pi(stats))
        return mask.astype(bool), stats

    def prune_sharded(self, max_rmsd, max_dev, min_per_group=20, row_block=0):
        """fc_prune_rmsd_sharded: this rank's part of the prune over the communicator of
        ``firecode_amd.dist.comm_init`` (a group of one without it) -> (mask, stats)."""
        mask = np.zeros(self.N, dtype=np.uint8)
        stats = np.zeros(6, dtype=np.int64)
        call("fc_prune_rmsd_sharded", self.handle, float(max_rmsd), float(max_dev), int(min_per_group),
             int(row_block), pb(mask), pi(stats))
        return mask.astype(bool), stats

    def bench_prune_sharded(self, max_rmsd, max_dev, reps=1, overlap=True):
        """``reps`` stream-ordered sharded prunes -> (screen kernel ms, step ms, mask, stats (8))."""
        mask = np.zeros(self.N, dtype=np.uint8)
        stats = np.zeros(8, dtype=np.int64)
        t_k, t_s = C.c_double(0), C.c_double(0)
        call("fc_bench_prune_rmsd_sharded", self.handle, float(max_rmsd), float(max_dev), int(reps),
             int(bool(overlap)), C.byref(t_k), C.byref(t_s), pb(mask), pi(stats))
        return t_k.value, t_s.value, mask.astype(bool), stats

    def bench_prune(self, max_rmsd, max_dev, reps=1, want_mask=True):
        mask = np.zeros(self.N, dtype=np.uint8) if want_mask else None
        stats = np.zeros(8, dtype=np.int64)  # [6], [7]: the subset stage of the lean fp32 screen
        t_k, t_s = C.c_double(0), C.c_double(0)
        call("fc_bench_prune_rmsd", self.handle, float(max_rmsd), float(max_dev), int(reps),
             C.byref(t_k), C.byref(t_s), pb(mask), pi(stats))
        return t_k.value, t_s.value, (None if mask is None else mask.astype(bool)), stats


class _EnsembleView(DeviceEnsemble):
    """A handle owned by another ensemble (its twin workspace): never destroyed from here."""

    def __init__(self, parent, handle):
        self._parent = parent
        self._view_h = handle
        self.N, self.A_all, self.W = parent.N, parent.A_all, parent.W

    @property
    def handle(self):
        self._parent.handle  # raises when the owner is gone
        return self._view_h

    def close(self):
        pass

    def twin(self):
        return self._parent


def prune_many(ensembles, max_rmsd, max_dev, min_per_group=20):
    """fc_prune_rmsd_many over DeviceEnsemble objects: list of bool masks, survivor counts."""
    n = len(ensembles)
    handles = (_ens * max(n, 1))(*[e.handle for e in ensembles])
    masks = [np.zeros(max(int(e.N), 1), dtype=np.uint8) for e in ensembles]
    ptrs = (_p_u8 * max(n, 1))(*[m.ctypes.data_as(_p_u8) for m in masks])
    survivors = np.zeros(max(n, 1), dtype=np.int64)
    call("fc_prune_rmsd_many", handles, n, float(max_rmsd), float(max_dev), int(min_per_group), ptrs,
         survivors.ctypes.data_as(_p_i64))
    return [m[: int(e.N)].astype(bool) for m, e in zip(masks, ensembles)], survivors[:n]


def unpack_bits(bits, n):
    """(rows, W) uint64 words -> (rows, n) bool."""
    b = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")
    return b[:, :n].astype(bool)


# ---- .xyz wire format (host code of the library) --------------------------------------
def xyz_write(path, atoms, coords, label="", mode=0):
    """mode 0: Ensemble.to_xyz text (label = basename); mode 1: utils.write_xyz text per conformer."""
    X = f64(coords)
    if X.ndim == 2:
        X = X[None]
    syms = [str(a).encode() for a in atoms]
    arr = (C.c_char_p * len(syms))(*syms)
    call("fc_xyz_write", os.fsencode(str(path)), arr, len(syms), pf(X), X.shape[0], str(label).encode(), int(mode))


def xyz_read(path):
    """-> (atoms (A,) str array of the first conformer, coords (N, A, 3))"""
    n, a = C.c_int64(0), C.c_int64(0)
    p = os.fsencode(str(path))
    call("fc_xyz_scan", p, C.byref(n), C.byref(a))
    coords = np.empty((n.value, a.value, 3))
    buf = C.create_string_buffer(max(a.value, 1) * 8)
    if n.value:
        call("fc_xyz_read", p, n.value, a.value, buf, pf(coords))
    atoms = np.array([buf.raw[i * 8:(i + 1) * 8].split(b"\0", 1)[0].decode() for i in range(a.value)])
    return atoms, coords


class _PinnedBlock:
    """A block of page-locked host memory (fc_host_alloc_pinned) exposed through the array interface; freed with the last
    array that views it."""

    def __init__(self, nbytes):
        ptr = C.c_void_p()
        call("fc_host_alloc_pinned", int(nbytes), C.byref(ptr))
        self._ptr = ptr.value
        self.__array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (self._ptr, False), "version": 3}

    def __del__(self):
        p, self._ptr = getattr(self, "_ptr", None), None
        if p:
            try:
                load().fc_host_free_pinned(C.c_void_p(p))
            except Exception:  # interpreter shutdown
                pass


def pinned_empty(shape, dtype=np.float64):
    """``np.empty(shape, dtype)`` in page-locked host memory: an ensemble built into such an array is uploaded by direct DMA
    (``prune_by_rmsd`` and every other entry point take it like any other array; host arrays in -> mask out 0.78 -> 0.6 ms
    at 10 000 x 50).  The memory is released with the last array that views it."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n == 0:
        return np.empty(shape, dtype)
    return np.asarray(_PinnedBlock(n)).view(dtype).reshape(shape)
