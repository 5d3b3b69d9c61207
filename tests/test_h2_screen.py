"""The split-half screen (kind 16: single-precision covariance on the f16 matrix pipe): what its error
bounds assume about the hardware, the covariances it accumulates against fp64, and its results against
the fp64 screen on shapes that exercise every template instance (1..4 k-steps of 32 atoms), ragged sizes,
uncentred and badly scaled coordinates.  The screen is a filter in front of the exact fp64 refine: the
bar is that it never drops a similar pair (bits, grey counts and masks equal the fp64 screen's)."""

import ctypes as C

import numpy as np
import pytest

from firecode_amd import synthetic as syn

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


def test_f16_matrix_pipe_model(fc):
    """the facts kabsch_h2_bounds (fc_kabsch_math.h) rests on, on the device the tests run on"""
    from firecode_amd import _lib

    flags = np.zeros(8, dtype=np.int64)
    worst = C.c_double(-1.0)
    _lib.call("fc_debug_mfma_f16_model", 20000, _lib.pi(flags), C.byref(worst))
    assert flags.tolist() == [1] * 8, flags
    # the bounds charge 66 u (|C| + sum |a b|) per instruction (order-independent); the library refuses the screen above 18 u
    assert 0.0 <= worst.value <= 18.0, worst.value


def _cov_tile(fc, ens, ib, jb):
    from firecode_amd import _lib

    B = np.zeros(256 * 9, dtype=np.float32)
    scale, bound = C.c_double(0.0), C.c_double(0.0)
    _lib.call("fc_debug_h2_covariance", ens.handle, int(ib), int(jb), B.ctypes.data_as(C.POINTER(C.c_float)),
              C.byref(scale), C.byref(bound))
    return B.reshape(16, 16, 3, 3).astype(np.float64), scale.value, bound.value


@pytest.mark.parametrize("n,a,seed,kind", [(320, 50, 71, "bench"), (96, 13, 72, "bench"), (128, 90, 73, "bench"),
                                            (64, 128, 74, "bench"), (160, 40, 75, "offset"), (160, 33, 76, "tiny"),
                                            (160, 64, 77, "huge"), (160, 50, 78, "spread")])
def test_h2_covariance_within_its_bound(fc, n, a, seed, kind):
    """accumulators of the screen (same instructions, same order) against the fp64 covariance: the entry error
    in units of s = (Gp + Gq)/2 stays inside the bound the polynomial bounds start from"""
    rng = np.random.default_rng(seed)
    X, _, _ = syn.synthetic_ensemble(n, a, seed=seed)
    center = True
    if kind == "offset":  # uncentred, 60 A from the origin
        X = X + np.array([60.0, -35.0, 20.0])
        center = False
    elif kind == "tiny":  # coordinates of 1e-4 A
        X = X * 1e-4
    elif kind == "huge":  # coordinates of 1e4 A
        X = X * 1e4
    elif kind == "spread":  # a few conformers 40 times smaller than the others, some atoms at the centroid
        X[::7] *= 0.025
        X[:, :3, :] = X.mean(axis=1, keepdims=True) + 1e-7 * rng.normal(size=(n, 3, 3))
    Xc = X - X.mean(axis=1, keepdims=True) if center else X
    G = (Xc ** 2).sum(axis=(1, 2))
    worst = 0.0
    with fc.DeviceEnsemble(X, center=center) as ens:
        for ib, jb in [(0, 0), (0, 16), (16, 48), ((n // 16 - 1) * 16, (n // 16 - 1) * 16), (32, 0)]:
            B, scale, bound = _cov_tile(fc, ens, ib, jb)
            assert scale > 0.0 and bound <= (13.02 + 66.4 * ((a + 31) // 32)) * U * (1 + 1e-12)  # kabsch_h2_entry_bound
            assert 2.0 ** 24 <= G.max() * scale * scale <= 2.0 ** 26 * (1 + 1e-12)
            ref = np.einsum("iax,jay->ijxy", Xc[ib:ib + 16], Xc[jb:jb + 16])
            s = 0.5 * (G[ib:ib + 16, None] + G[None, jb:jb + 16])
            err = np.abs(B / (scale * scale) - ref).max(axis=(2, 3)) / s
            # pairs below the kernel's floor (scaled s < A: their subnormal-lo term is not covered) go to the exact path
            covered = s * scale * scale >= a
            assert covered.any()
            worst = max(worst, float((err[covered] / bound).max()))
    assert worst <= 0.5, worst  # observed: a few per cent of the worst-case bound


@pytest.mark.parametrize("n,a,seed,thr", [(300, 50, 81, 0.5), (257, 13, 82, 0.35), (200, 90, 83, 0.25), (130, 128, 84, 0.4),
                                          (190, 33, 85, 1.2), (100, 64, 86, 0.5)])
def test_h2_screen_equals_fp64_screen(fc, n, a, seed, thr):
    from firecode_amd import _lib
    from firecode_amd._lib import unpack_bits

    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
    out = {}
    try:
        for kind in (64, 16):
            _lib.screen_select(kind)
            with fc.DeviceEnsemble(X, center=True) as ens:
                bits, grey = ens.simbits(thr, 2 * thr)
                # (beyond ~100 atoms the fp64 column tile does not fit LDS: "64" is then the fp64 VALU screen, kind 1)
                assert _lib.screen_last_kind() in ((16,) if kind == 16 else (64, 1))
                mask, stats = ens.prune(thr, 2 * thr)
                assert _lib.screen_last_kind() in ((16,) if kind == 16 else (64, 1))
            out[kind] = (unpack_bits(bits, n), grey, mask, stats[:3].tolist())
    finally:
        _lib.screen_select(0)
    assert np.array_equal(out[64][0], out[16][0]) and out[64][1] == out[16][1]
    assert np.array_equal(out[64][2], out[16][2]) and out[64][3][2] == out[16][3][2]
    assert out[16][0].any()


def test_h2_screen_badly_scaled_and_degenerate_inputs(fc):
    """huge / tiny coordinates, an ensemble far from the origin, duplicate and collapsed conformers, NaN:
    the split-half screen (forced) gives the fp64 screen's bits, or declines (FC_E_INVALID) and never lies"""
    from firecode_amd import _lib
    from firecode_amd._lib import unpack_bits

    X0, _, _ = syn.synthetic_ensemble(192, 40, seed=91)
    cases = {
        "huge": (X0 * 3e3, 0.5 * 3e3), "tiny": (X0 * 1e-3, 0.5e-3), "offset": (X0 + np.array([300.0, 0.0, -100.0]), 0.5),
        "duplicates": (np.concatenate([X0[:96], X0[:96]]), 0.5),
    }
    collapsed = X0.copy()
    collapsed[5] = collapsed[5].mean(axis=0)  # all atoms on one point: G = 0 after centring
    collapsed[9] = collapsed[9].mean(axis=0)
    cases["collapsed"] = (collapsed, 0.5)
    for name, (X, thr) in cases.items():
        out = {}
        try:
            for kind in (64, 16):
                _lib.screen_select(kind)
                with fc.DeviceEnsemble(X, center=(name != "offset")) as ens:
                    bits, grey = ens.simbits(thr, 2 * thr)
                    mask, _ = ens.prune(thr, 2 * thr)
                out[kind] = (unpack_bits(bits, len(X)), grey, mask)
        finally:
            _lib.screen_select(0)
        assert np.array_equal(out[64][0], out[16][0]) and out[64][1] == out[16][1], name
        assert np.array_equal(out[64][2], out[16][2]), name
    # NaN coordinates: whatever the screen, the pairs of that conformer reach the exact path and are not similar
    Xn = X0.copy()
    Xn[7, 3, 1] = np.nan
    res = {}
    try:
        for kind in (64, 16, 0):
            _lib.screen_select(kind)
            try:
                with fc.DeviceEnsemble(Xn, center=True) as ens:
                    bits, _ = ens.simbits(0.5, 1.0)
                res[kind] = unpack_bits(bits, len(Xn))
            except fc.FirecodeHipInputError:
                assert kind == 16  # a NaN norm leaves no scale to take: the forced split-half screen declines
    finally:
        _lib.screen_select(0)
    assert np.array_equal(res[64], res[0])
    if 16 in res:
        assert np.array_equal(res[64], res[16])


@pytest.mark.parametrize("a", [1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 96, 97, 127, 128])
def test_prune_masks_over_ragged_sizes_and_atom_counts(fc, a):
    """prune_by_rmsd (default screen: split-half up to 128 atoms) against the oracle for conformer counts around
    the tile edges (1 ... 193) and atom counts around the 32-atom k-steps"""
    from oracle import cpu_ref as o

    rng = np.random.default_rng(1000 + a)
    atoms = np.array(["C"] * a)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 193):
        k = max(1, n // 3)
        base = rng.normal(scale=2.0, size=(k, a, 3))
        X = base[rng.integers(0, k, n)] + rng.normal(scale=0.02, size=(n, a, 3))
        thr = 0.3
        _, m = fc.pruner.prune_by_rmsd(X, atoms, thr)
        _, ref = o.prune_by_rmsd(X, atoms, thr)
        assert np.array_equal(m, ref), (a, n)
