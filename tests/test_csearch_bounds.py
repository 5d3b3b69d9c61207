"""The two error bounds the round-5 csearch kernels lean on, checked in NumPy (no GPU):

* 16-bit angles of the TFD first match (csrc/fc_prune.hip: k_tfd_pack_u16 and the kernels behind it): with
  u = rint(a * 65536 / 360) mod 65536 the sum of the wrapping 16-bit distances differs from the reference's deviation
  (firecode/torsion_module.py:1056-1067: sum of |d| or |d - 360|) times 65536 / 360 by at most one unit per angle;
* the closed form of a pair distance under the back-off rotations of the scan (csrc/fc_torsion.hip: torsion_step):
  c0 + c1 cos(b d) + c2 sin(b d) against the distance after b walked rotations (oracle.rotate_dihedral, the
  reference's rotate_dihedral restated), within 4e-14 (|R|^2 + |P|^2 + 1) over 60 steps -- the kernel's guard is 1e-10."""
import numpy as np
import pytest

from oracle import cpu_ref as o

S16 = 65536.0 / 360.0


def _u16(a):
    return np.rint(a * S16).astype(np.int64) & 0xFFFF


def _sad_u16(ua, ub):
    d = (ub - ua) & 0xFFFF
    d = np.where(d >= 32768, 65536 - d, d)  # the wrapping distance: |x - 0x8000| of the biased difference in the kernel
    return d.sum(axis=-1)


@pytest.mark.parametrize("q", [1, 3, 8])
def test_first_match_16bit_sum_within_one_unit_per_angle(q):
    rng = np.random.default_rng(40 + q)
    a = rng.uniform(-270, 270, size=(200000, q))
    b = a + rng.uniform(-12, 12, size=a.shape) * (rng.random(a.shape) < 0.7)
    b = np.where(np.abs(b) <= 270, b, a)  # the kernels refuse angles beyond +-270 (the delta is not circular there)
    # pairs across the wrap
    a[:1000, 0] = rng.uniform(175, 180, 1000)
    b[:1000, 0] = rng.uniform(-180, -175, 1000)
    d = np.abs(a - b)
    ref = np.abs(d - (d > 180) * 360).sum(axis=1)  # torsion_module.py:1063-1065
    sad = _sad_u16(_u16(a), _u16(b))
    assert np.abs(sad - ref * S16).max() <= q * 1.0 + 1e-6
    # the kernels' two verdicts around a threshold never contradict the exact comparison
    for thr in (10.0, 0.3, 40.0):
        ts = thr * S16
        t_lo, t_hi = max(0, int(np.floor(ts)) - 10), int(np.ceil(ts)) + 10
        assert (ref[sad < t_lo] < thr).all()
        assert (ref[sad >= t_hi] >= thr).all()


def test_back_off_distance_closed_form_against_walked_rotations():
    rng = np.random.default_rng(7)
    worst = 0.0
    for trial in range(60):
        n = 12
        x = rng.normal(size=(n, 3)) * rng.uniform(2, 15)
        torsion = (0, 1, 2, 3)
        mask = np.zeros(n, dtype=bool)
        mask[3:8] = True  # atoms 3..7 turn about the 1-2 bond, 8..11 rest
        backoff = int(rng.choice([2, 5, 7]))
        c = x[2].copy()
        axis = x[1] - x[2]
        nrm = axis / np.linalg.norm(axis)
        R = x[8:] - c          # resting atoms
        P = x[3:8] - c         # moving atoms
        Rn, Pn = R @ nrm, P @ nrm
        S = (R * R).sum(1)[:, None] + (P * P).sum(1)[None, :]
        c0 = S - 2.0 * Rn[:, None] * Pn[None, :]
        c1 = -2.0 * (R @ P.T - Rn[:, None] * Pn[None, :])
        c2 = 2.0 * (R @ np.cross(nrm, P).T)
        temp = x.copy()
        for b in range(1, 61):
            temp = o.rotate_dihedral(temp, torsion, -backoff, mask)
            walked = ((temp[8:, None, :] - temp[None, 3:8, :]) ** 2).sum(-1)
            ang = np.deg2rad(b * backoff)
            closed = c0 + c1 * np.cos(ang) + c2 * np.sin(ang)
            worst = max(worst, (np.abs(walked - closed) / (S + 1.0)).max())
    assert worst < 4e-13, worst  # (4e-14 seen; the kernel hands a pair within 1e-10 (S + 1) of the threshold to the walked loop)
