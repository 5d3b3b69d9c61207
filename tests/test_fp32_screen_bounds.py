"""Numerical check of the bounds behind the single-precision screen (csrc/fc_kabsch_math.h,
kabsch_f32_bounds / kabsch_may_be_below_f32): the kernel drops a pair only when the fp32 values
of P, P', P'' clear a bound on their own error.  Here the same arithmetic is replayed in NumPy
float32 -- inputs rounded to fp32, covariance accumulated atom by atom in fp32, the polynomial
in fp32 in the kernel's order -- and compared with float64 on the exact inputs.  No GPU needed:
this pins the CONSTANTS of the analysis (an error of a factor of a few would show here), the
GPU tests pin the kernel's results."""

import numpy as np
import pytest

U = 2.0 ** -24


def bounds(a4):  # kabsch_f32_bounds without its factor 2
    db = (a4 + 4.0) * U + U
    return 45.0 * db + 172.0 * U, 9.5 * db + 46.0 * U, 6.0 * db + 52.0 * U


def poly(B, s, L, dt):
    """P0, P1/4-like and P2/4-like values of kabsch_may_be_below(_f32) in units of s (dtype dt)."""
    r = (dt(1.0) / s.astype(dt)).astype(dt)
    b = (B.astype(dt) * r[:, None, None]).astype(dt)
    lq = (L.astype(dt) * r).astype(dt)
    n2 = (b * b).sum(axis=(1, 2), dtype=dt)
    L2 = lq * lq
    uu = L2 - n2
    c = np.empty_like(b)
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[:, i, j] = b[:, i1, j1] * b[:, i2, j2] - b[:, i1, j2] * b[:, i2, j1]
    det = (b[:, 0, :] * c[:, 0, :]).sum(axis=1, dtype=dt)
    e2 = (c * c).sum(axis=(1, 2), dtype=dt)
    P2 = dt(2.0) * L2 + uu
    P1 = uu * lq - dt(2.0) * det
    P0 = uu * uu - dt(4.0) * (e2 + dt(2.0) * lq * det)
    return P0, P1, P2


@pytest.mark.parametrize("A,offset,thr", [(13, 0.0, 0.5), (50, 0.0, 0.5), (50, 0.0, 0.25), (88, 0.0, 0.7), (50, 6.0, 0.5)])
def test_fp32_polynomial_stays_inside_the_proven_bounds(A, offset, thr):
    rng = np.random.default_rng(A)
    P = 4000
    base = rng.normal(scale=2.5, size=(P, A, 3))
    # pairs at every distance: copies with noise from 1e-3 to 2 A, half of them rotated
    noise = rng.normal(size=(P, A, 3)) * np.geomspace(1e-3, 2.0, P)[:, None, None]
    x = base - base.mean(axis=1, keepdims=True) + offset
    y = base + noise
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a, b, c, d = q.T
    R = np.stack([np.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)], -1),
                  np.stack([2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)], -1),
                  np.stack([2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1)], 1)
    y[::2] = np.einsum("pij,paj->pai", R[::2], y[::2])
    y = y - y.mean(axis=1, keepdims=True) + offset
    # exact side
    B64 = np.einsum("pax,pay->pxy", x, y)
    G = (x * x).sum(axis=(1, 2)) + (y * y).sum(axis=(1, 2))
    s = 0.5 * G
    L = s - 0.5 * A * thr * thr
    keep = (A * thr * thr) < 0.5 * s  # the kernel's "tiny" pairs are never screened
    assert keep.mean() > 0.9
    ref = poly(B64, s, L, np.float64)
    # fp32 side: inputs rounded, sequential accumulation, products rounded before they are added
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    B32 = np.zeros((P, 3, 3), dtype=np.float32)
    for k in range(A):
        B32 += xf[:, k, :, None] * yf[:, k, None, :]
    s32 = ((0.5 * (x * x).sum(axis=(1, 2))).astype(np.float32).astype(np.float64)
           + (0.5 * (y * y).sum(axis=(1, 2))).astype(np.float32).astype(np.float64))
    got = poly(B32, s32, (s32 - 0.5 * A * thr * thr), np.float32)
    a4 = (A + 3) // 4 * 4
    for name, g, r, bd in zip(("P0", "P1", "P2"), got, ref, bounds(a4)):
        err = np.abs(g.astype(np.float64) - r)[keep]
        assert err.max() <= bd, (name, err.max(), bd)
        # and the analysis is not absurdly loose either: within 4 orders of magnitude of what happens
        assert err.max() > bd * 1e-4, (name, err.max(), bd)


def test_decisions_are_conservative():
    """a pair whose exact msd is below the threshold is never dropped by the bounded fp32 test"""
    rng = np.random.default_rng(3)
    A, P, thr = 40, 20000, 0.5
    base = rng.normal(scale=2.0, size=(P, A, 3))
    x = base - base.mean(axis=1, keepdims=True)
    scale = np.concatenate([np.full(P // 2, 1.0), np.geomspace(0.2, 3.0, P - P // 2)])
    y = base + rng.normal(size=(P, A, 3)) * (thr / np.sqrt(3.0)) * scale[:, None, None]
    y = y - y.mean(axis=1, keepdims=True)
    B64 = np.einsum("pax,pay->pxy", x, y)
    G = (x * x).sum(axis=(1, 2)) + (y * y).sum(axis=(1, 2))
    sv = np.linalg.svd(B64, compute_uv=False)
    sv[:, 2] *= np.sign(np.linalg.det(B64))
    msd = (G - 2.0 * sv.sum(axis=1)) / A
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    B32 = np.zeros((P, 3, 3), dtype=np.float32)
    for k in range(A):
        B32 += xf[:, k, :, None] * yf[:, k, None, :]
    s = 0.5 * G
    P0, P1, P2 = poly(B32, s, s - 0.5 * A * thr * thr, np.float32)
    b0, b1, b2 = (2.0 * v for v in bounds(40))
    dropped = (P2 > b2) & (P1 > b1) & (P0 > b0)
    similar = msd < thr * thr
    assert similar.sum() > 1000 and (~similar).sum() > 1000
    assert not (dropped & similar).any()
    # the band: dissimilar pairs the fp32 test keeps are all close to the threshold
    kept_far = (~dropped) & (msd > thr * thr + 0.05)
    assert not kept_far.any()


@pytest.mark.parametrize("scale", [1.0, 1e-3, 300.0])
def test_unscaled_form_of_the_kernel(scale):
    """the kernel compares P'', P', P with p2 s^2, p1 s^3, p0 s^4 instead of scaling B and L by 1/s:
    the same decisions as the scaled form for structures measured in Angstrom, in nm-like and in
    pm-like units (fp32 range: s^4 up to ~1e38)"""
    rng = np.random.default_rng(9)
    A, P, thr = 50, 6000, 0.5 * scale
    base = rng.normal(scale=2.5 * scale, size=(P, A, 3))
    x = base - base.mean(axis=1, keepdims=True)
    y = base + rng.normal(size=(P, A, 3)) * (np.geomspace(0.02, 2.0, P) * scale)[:, None, None]
    y = y - y.mean(axis=1, keepdims=True)
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    B = np.zeros((P, 3, 3), dtype=np.float32)
    for k in range(A):
        B += xf[:, k, :, None] * yf[:, k, None, :]
    f = np.float32
    s = ((0.5 * (x * x).sum(axis=(1, 2))).astype(f) + (0.5 * (y * y).sum(axis=(1, 2))).astype(f)).astype(f)
    L = (s - f(0.5 * A * thr * thr)).astype(f)
    n2 = (B * B).sum(axis=(1, 2), dtype=f)
    uu = L * L - n2
    c = np.empty_like(B)
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[:, i, j] = B[:, i1, j1] * B[:, i2, j2] - B[:, i1, j2] * B[:, i2, j1]
    det = (B[:, 0, :] * c[:, 0, :]).sum(axis=1, dtype=f)
    e2 = (c * c).sum(axis=(1, 2), dtype=f)
    P2, P1, P0 = f(2) * L * L + uu, uu * L - f(2) * det, uu * uu - f(4) * (e2 + f(2) * L * det)
    b0, b1, b2 = (f(2.0 * v) for v in bounds(52))
    s2 = s * s
    dropped = (P2 > b2 * s2) & (P1 > b1 * s2 * s) & (P0 > b0 * s2 * s2)
    assert np.isfinite(P0).all() and np.isfinite(s2 * s2).all()
    sv = np.linalg.svd(np.einsum("pax,pay->pxy", x, y), compute_uv=False)
    sv[:, 2] *= np.sign(np.linalg.det(np.einsum("pax,pay->pxy", x, y)))
    msd = ((x * x).sum(axis=(1, 2)) + (y * y).sum(axis=(1, 2)) - 2.0 * sv.sum(axis=1)) / A
    similar = msd < thr * thr
    assert similar.sum() > 300 and dropped.sum() > 300
    assert not (dropped & similar).any()
    assert not ((~dropped) & (msd > 1.2 * thr * thr)).any()  # the band stays narrow in every unit system


@pytest.mark.parametrize("A,shape,thr", [(50, "ball", 0.5), (50, "chain", 0.5), (80, "chain", 0.35), (23, "ball", 0.7)])
def test_subset_stage_is_conservative_and_inside_its_bounds(A, shape, thr):
    """The subset stage of the lean fp32 screen (k_simbits_screen_mfma_f32<.., STAGED>): atoms of the even
    groups of four only, accumulators hold UNcentred sums, the rank-one centring term C_S(p) C_S(q)^T / A_S
    is subtracted in fp32, the polynomial is compared with bounds scaled by the uncentred norms.  Replayed
    in NumPy float32: (1) the fp32 values of P, P', P'' stay within the bounds the kernel uses (in units of
    the uncentred scale), (2) a pair whose FULL msd is below the threshold is never dropped -- chain-like
    structures included, where the subset's centroid sits far from the structure's."""
    rng = np.random.default_rng(A + len(shape))
    P = 6000
    if shape == "ball":
        base = rng.normal(scale=2.5, size=(P, A, 3))
    else:  # a random walk: elongated, subset centroid well away from the full centroid
        base = np.cumsum(rng.normal(scale=0.9, size=(P, A, 3)), axis=1)
    x = base - base.mean(axis=1, keepdims=True)
    y = base + rng.normal(size=(P, A, 3)) * np.concatenate([np.full(P // 2, thr / np.sqrt(3.0) * 0.9),
                                                              np.geomspace(0.05, 3.0, P - P // 2)])[:, None, None]
    y = y - y.mean(axis=1, keepdims=True)
    sub = ((np.arange(A) >> 2) & 1) == 0
    A_S, A4_S = int(sub.sum()), 4 * (((A + 3) // 4 + 1) // 2)
    f = np.float32
    # exact side: full msd, and the subset quantities the stage tests
    B = np.einsum("pax,pay->pxy", x, y)
    sv = np.linalg.svd(B, compute_uv=False)
    sv[:, 2] *= np.sign(np.linalg.det(B))
    msd_full = ((x * x).sum(axis=(1, 2)) + (y * y).sum(axis=(1, 2)) - 2.0 * sv.sum(axis=1)) / A
    xs, ys = x[:, sub], y[:, sub]
    Cx, Cy = xs.sum(axis=1), ys.sum(axis=1)
    Bc = np.einsum("pax,pay->pxy", xs, ys) - Cx[:, :, None] * Cy[:, None, :] / A_S
    Gc = ((xs * xs).sum(axis=(1, 2)) - (Cx * Cx).sum(axis=1) / A_S) + ((ys * ys).sum(axis=(1, 2)) - (Cy * Cy).sum(axis=1) / A_S)
    Gu = (xs * xs).sum(axis=(1, 2)) + (ys * ys).sum(axis=(1, 2))
    s_c, s_u = 0.5 * Gc, 0.5 * Gu
    L = s_c - 0.5 * A * thr * thr
    # the lower bound itself (exact arithmetic): the subset residual never exceeds the full one
    svs = np.linalg.svd(Bc, compute_uv=False)
    svs[:, 2] *= np.sign(np.linalg.det(Bc))
    assert np.all(Gc - 2.0 * svs.sum(axis=1) <= A * msd_full * (1 + 1e-12) + 1e-9)
    # fp32 side, as the kernel does it
    xf, yf = xs.astype(f), ys.astype(f)
    acc = np.zeros((P, 3, 3), dtype=f)
    for k in range(A_S):
        acc += xf[:, k, :, None] * yf[:, k, None, :]
    inv = f(1.0 / A_S)
    cp, cq = Cx.astype(f), Cy.astype(f)
    B32 = (acc + (-(inv * cp))[:, :, None] * cq[:, None, :]).astype(f)
    half = lambda v: (0.5 * v).astype(f)  # noqa: E731  G_S^c / 2 and G_S^u / 2 are stored per conformer
    gxc = half((xs * xs).sum(axis=(1, 2)) - (Cx * Cx).sum(axis=1) / A_S)
    gyc = half((ys * ys).sum(axis=(1, 2)) - (Cy * Cy).sum(axis=1) / A_S)
    gxu, gyu = half((xs * xs).sum(axis=(1, 2))), half((ys * ys).sum(axis=(1, 2)))
    s32, su32 = (gxc + gyc).astype(f), (gxu + gyu).astype(f)
    L32 = (s32 - f(0.5 * A * thr * thr)).astype(f)
    n2 = (B32 * B32).sum(axis=(1, 2), dtype=f)
    uu = L32 * L32 - n2
    c = np.empty_like(B32)
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[:, i, j] = B32[:, i1, j1] * B32[:, i2, j2] - B32[:, i1, j2] * B32[:, i2, j1]
    det = (B32[:, 0, :] * c[:, 0, :]).sum(axis=1, dtype=f)
    e2 = (c * c).sum(axis=(1, 2), dtype=f)
    P2, P1, P0 = f(2) * L32 * L32 + uu, uu * L32 - f(2) * det, uu * uu - f(4) * (e2 + f(2) * L32 * det)
    b0, b1, b2 = (f(2.0 * v) for v in bounds(A4_S + 8))   # kabsch_f32_bounds(4 KS1 + 8), factor 2 as in the launcher
    su2 = su32 * su32
    tiny = ~(f(4.0) * f(0.5 * A * thr * thr) < s32)
    dropped = ~tiny & (P2 > b2 * su2) & (P1 > b1 * su2 * su32) & (P0 > b0 * su2 * su2)
    similar = msd_full < thr * thr
    assert similar.sum() > 500 and dropped.sum() > 500
    assert not (dropped & similar).any()                      # (2) conservative
    # (1) errors against the exact subset polynomial, in units of the UNcentred scale
    ref = poly(Bc, s_u, L, np.float64)
    got = poly(B32, su32.astype(np.float64), L32.astype(np.float64), np.float64)  # same fp32 inputs, exact polynomial
    P0u, P1u, P2u = (P0.astype(np.float64) / s_u**4, P1.astype(np.float64) / s_u**3, P2.astype(np.float64) / s_u**2)
    for name, g, r, bd in zip(("P0", "P1", "P2"), (P0u, P1u, P2u), ref, bounds(A4_S + 8)):
        err = np.abs(g - r)[~tiny]
        assert err.max() <= bd, (name, err.max(), bd)
    del got


# ---------------------------------------------------------------------------------------------
# the split-half screen (k_simbits_screen_mfma_h2): its two-test polynomial and its operand split
# ---------------------------------------------------------------------------------------------
def _lambda_max(B):
    """largest eigenvalue of Horn's quaternion matrix of each 3 x 3 covariance (float64, LAPACK)"""
    K = np.empty((len(B), 4, 4))
    Sxx, Sxy, Sxz = B[:, 0, 0], B[:, 0, 1], B[:, 0, 2]
    Syx, Syy, Syz = B[:, 1, 0], B[:, 1, 1], B[:, 1, 2]
    Szx, Szy, Szz = B[:, 2, 0], B[:, 2, 1], B[:, 2, 2]
    K[:, 0, 0] = Sxx + Syy + Szz
    K[:, 0, 1] = K[:, 1, 0] = Syz - Szy
    K[:, 0, 2] = K[:, 2, 0] = Szx - Sxz
    K[:, 0, 3] = K[:, 3, 0] = Sxy - Syx
    K[:, 1, 1] = Sxx - Syy - Szz
    K[:, 1, 2] = K[:, 2, 1] = Sxy + Syx
    K[:, 1, 3] = K[:, 3, 1] = Szx + Sxz
    K[:, 2, 2] = -Sxx + Syy - Szz
    K[:, 2, 3] = K[:, 3, 2] = Syz + Szy
    K[:, 3, 3] = -Sxx - Syy + Szz
    return np.linalg.eigvalsh(K)


def test_two_sign_tests_decide_like_three():
    """kabsch_may_be_below_f32_2t: u = L^2 - |B|_F^2 > 0 and P(L) > 0 prove lambda_max < L (three roots are below
    |B|_F then); checked in float64 on random, rank-deficient, nearly collinear and reflected covariances,
    with L swept across every root"""
    rng = np.random.default_rng(123)
    n = 40000
    B = rng.normal(size=(n, 3, 3))
    B[: n // 4] *= np.array([1.0, 0.3, 0.02])[None, None, :]           # anisotropic
    u1, v1 = rng.normal(size=(2, n // 8, 3))
    B[n // 4: n // 4 + n // 8] = u1[:, :, None] * v1[:, None, :] + 1e-3 * rng.normal(size=(n // 8, 3, 3))  # nearly rank one
    B[n // 2: n // 2 + n // 8, :, 2] *= -1.0                            # reflections
    ev = _lambda_max(B)                                                 # ascending: ev[:, 3] is lambda_max
    n2 = (B * B).sum(axis=(1, 2))
    # second-largest root never exceeds the Frobenius norm (what the two-test form rests on)
    assert np.all(ev[:, 2] <= np.sqrt(n2) * (1 + 1e-12))
    for f in (0.2, 0.7, 0.95, 0.999, 1.001, 1.05, 1.5, 3.0):
        for ref in (ev[:, 3], ev[:, 2], np.sqrt(n2)):
            L = np.abs(ref) * f + 1e-9
            s = np.ones(n)
            P0, P1, P2 = poly(B, s, L, np.float64)
            uu = L * L - n2
            proven_2t = (uu > 0) & (P0 > 0)
            proven_3t = (P0 > 0) & (P1 > 0) & (P2 > 0)
            truth = ev[:, 3] < L
            margin = np.abs(ev[:, 3] - L) > 1e-9 * np.maximum(1.0, np.abs(L))  # away from the root itself
            assert not np.any(proven_2t & ~truth & margin)   # never claims a similar pair dissimilar
            assert not np.any(proven_3t & ~truth & margin)
            # where u > 0 the two forms agree; u <= 0 the kernel hands to the three-test form
            assert np.array_equal((proven_2t & margin)[uu > 0], (proven_3t & margin)[uu > 0])


def test_split_half_representation_error():
    """x 2^e = hi + lo + d with |d| <= 2^-22 (1 + 2^-12) |x 2^e| or <= 2^-25 where lo is subnormal (k_f64_to_h2:
    conversions through float32 round twice), and the covariance of the split operands -- products exact, the
    lo lo^T term left out -- stays within (8.01 + 1 + 4.01) u s of the exact one: the representation part of
    kabsch_h2_entry_bound (the matrix pipe's own part is checked on the GPU)"""
    rng = np.random.default_rng(7)
    for scale_pow, spread in ((0, 1.0), (5, 1.0), (-9, 1.0), (0, 1e-3)):
        A = 50
        x = rng.normal(scale=2.5, size=(3000, A, 3)) * spread
        y = rng.normal(scale=2.5, size=(3000, A, 3))
        x -= x.mean(axis=1, keepdims=True)
        y -= y.mean(axis=1, keepdims=True)
        x *= 2.0 ** scale_pow
        y *= 2.0 ** scale_pow
        gmax = max((x * x).sum(axis=(1, 2)).max(), (y * y).sum(axis=(1, 2)).max())
        m, ex = np.frexp(np.sqrt(gmax))
        sc = 2.0 ** (13 - ex)

        def split(v):
            vs = v * sc
            hi = vs.astype(np.float32).astype(np.float16)
            lo = (vs - hi.astype(np.float64)).astype(np.float32).astype(np.float16)
            return vs, hi.astype(np.float64), lo.astype(np.float64)

        xs, xh, xl = split(x)
        ys, yh, yl = split(y)
        assert np.abs(xh).max() <= 2.0 ** 13 * (1 + 2.0 ** -11)
        d = np.abs(xs - xh - xl)
        assert np.all(d <= np.maximum(2.0 ** -22 * (1 + 2.0 ** -12) * np.abs(xs), 2.0 ** -25))
        B_exact = np.einsum("pax,pay->pxy", xs, ys)
        B_split = (np.einsum("pax,pay->pxy", xh, yh) + np.einsum("pax,pay->pxy", xh, yl) + np.einsum("pax,pay->pxy", xl, yh))
        s = 0.5 * ((xs * xs).sum(axis=(1, 2)) + (ys * ys).sum(axis=(1, 2)))
        covered = s >= A  # the kernel's tiny_floor
        err = np.abs(B_split - B_exact).max(axis=(1, 2)) / s
        assert covered.all() or spread < 1.0
        assert err[covered].max() <= (8.01 + 1.0 + 4.01) * U
