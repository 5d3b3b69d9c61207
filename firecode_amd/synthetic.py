"""Synthetic workloads of SURVEY.md section 8d / BASELINE.md section 3 (data
generation only -- no part of the compute path)."""

import numpy as np
from scipy.spatial.distance import cdist


def random_rotation(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def synthetic_skeleton(n_atoms, rng, bond=1.5, min_dist=1.2):
    """Self-avoiding random walk, bond 1.5 A, no atom closer than 1.2 A."""
    pts = [np.zeros(3)]
    while len(pts) < n_atoms:
        for _ in range(1000):
            d = rng.normal(size=3)
            cand = pts[-1] + bond * d / np.linalg.norm(d)
            if np.min(np.linalg.norm(np.array(pts) - cand, axis=1)) >= min_dist:
                pts.append(cand)
                break
        else:  # dead end: restart
            pts = [np.zeros(3)]
    return np.array(pts)


def compact_skeleton(n_atoms, rng, spacing=1.5):
    """A globule instead of a walk: the n_atoms points of a cubic lattice (spacing 1.5 A) nearest to the origin, jittered
    by 0.1 A -- the radius of gyration of a folded molecule or of docked poses (4.6 A at 260 atoms) where the
    self-avoiding walk of synthetic_skeleton gives 15-21 A."""
    m = int(np.ceil((n_atoms * 2) ** (1.0 / 3.0))) + 2
    g = np.arange(-m, m + 1) * spacing
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3) + spacing / 3.0
    pts = pts[np.argsort((pts ** 2).sum(axis=1), kind="stable")[:n_atoms]]
    return pts + rng.normal(scale=0.1, size=pts.shape)


def synthetic_ensemble(n_conf, n_atoms, seed, cluster_size=5, sigma_cluster=0.6, sigma_conf=0.03, compact=False):
    """K = n_conf / cluster_size cluster centres (skeleton + N(0, 0.6^2), no
    self-clash below 0.5 A), members = centre + N(0, 0.03^2), shuffled, each
    with a random proper rotation and a translation N(0, 5^2).  Intra-cluster
    RMSD ~0.07 A, inter-cluster ~1.5 A: no pair near the 0.5 A threshold.
    Returns (coords (N, A, 3), atoms (A,) all 'C', cluster id per conformer)."""
    rng = np.random.default_rng(seed)
    skel = compact_skeleton(n_atoms, rng) if compact else synthetic_skeleton(n_atoms, rng)
    K = max(1, n_conf // cluster_size)
    centres = np.empty((K, n_atoms, 3))
    for k in range(K):
        while True:
            c = skel + rng.normal(scale=sigma_cluster, size=skel.shape)
            if compact:  # (a globule's many 1.5 A neighbours never all survive the noise: no clash rejection there)
                centres[k] = c
                break
            d = cdist(c, c)
            d[np.diag_indices(n_atoms)] = 10.0
            if d.min() >= 0.5:
                centres[k] = c
                break
    assign = np.arange(n_conf) % K
    rng.shuffle(assign)
    coords = centres[assign] + rng.normal(scale=sigma_conf, size=(n_conf, n_atoms, 3))
    # random rigid motion per conformer (vectorised QR of Gaussian matrices)
    q, r = np.linalg.qr(rng.normal(size=(n_conf, 3, 3)))
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    neg = np.linalg.det(q) < 0
    q[neg, :, 0] *= -1.0
    coords = np.einsum("nij,naj->nai", q, coords) + rng.normal(scale=5.0, size=(n_conf, 1, 3))
    return np.ascontiguousarray(coords), np.array(["C"] * n_atoms), assign


def continuous_ensemble(n_conf, n_atoms, seed=11, n_modes=6, thr=0.5):
    """An ensemble WITHOUT cluster structure: one skeleton displaced along ``n_modes`` random
    orthonormal collective modes with Gaussian amplitudes.  Pair RMSDs spread continuously from 0
    to ~5 x ``thr``, ~1.5 % of the pairs fall below ``thr`` and the density is smooth across the
    threshold -- the case the clustered bench ensemble does not exercise (its pairs sit at 0.07 A
    or 1.5 A).  Returns coords (N, A, 3)."""
    rng = np.random.default_rng(seed)
    base, _, _ = synthetic_ensemble(1, n_atoms, seed=2)
    modes = np.linalg.qr(rng.normal(size=(n_atoms * 3, n_modes)))[0].T.reshape(n_modes, n_atoms, 3)
    amp = rng.normal(scale=0.35 * np.sqrt(n_atoms) * (thr / 0.5), size=(n_conf, n_modes))
    return np.ascontiguousarray(base[0][None] + np.einsum("nm,mac->nac", amp, modes))


def synthetic_trimolecular(n_conf=(2, 2, 2), n_atoms=(9, 11, 8), seed=0, pivots_per_conf=(1, 2, 1), sep=(3, 4, 3)):
    """Three small molecules for the trimolecular cyclical embed: per molecule ``coords``
    (n, A, 3) (conformers = jittered copies of a random skeleton), two reactive atoms
    ``sep`` bonds apart, and per conformer a list of pivots (start xyz, end xyz, start
    cumnum, end cumnum): orbital centres 0.8-1.2 A off the reactive atoms, as FIRECODE's
    Pivot objects hold them.  cumnum = atom index + atoms of the preceding molecules.
    Returns a list of three dicts (coords, reactive_indices, pivots, reactive_cumnums)."""
    rng = np.random.default_rng(seed)
    mols, offset = [], 0
    for m in range(3):
        A = n_atoms[m]
        skel = synthetic_skeleton(A, rng)
        r0 = int(rng.integers(0, A - sep[m]))
        r1 = r0 + sep[m]
        coords = skel[None] + rng.normal(scale=0.08, size=(n_conf[m], A, 3))
        coords = np.array([c @ random_rotation(rng).T + rng.normal(scale=2.0, size=3) for c in coords])
        pivots = []
        for c in range(n_conf[m]):
            centre = coords[c].mean(axis=0)
            plist = []
            for _ in range(pivots_per_conf[m]):
                off0 = coords[c, r0] - centre + rng.normal(scale=0.5, size=3)
                off1 = coords[c, r1] - centre + rng.normal(scale=0.5, size=3)
                start = coords[c, r0] + rng.uniform(0.8, 1.2) * off0 / np.linalg.norm(off0)
                end = coords[c, r1] + rng.uniform(0.8, 1.2) * off1 / np.linalg.norm(off1)
                plist.append((start, end, r0 + offset, r1 + offset))
            pivots.append(plist)
        mols.append({"coords": coords, "reactive_indices": np.array([r0, r1]), "pivots": pivots,
                     "reactive_cumnums": {r0: r0 + offset, r1: r1 + offset}})
        offset += A
    return mols
