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


def synthetic_ensemble(n_conf, n_atoms, seed, cluster_size=5, sigma_cluster=0.6, sigma_conf=0.03):
    """K = n_conf / cluster_size cluster centres (skeleton + N(0, 0.6^2), no
    self-clash below 0.5 A), members = centre + N(0, 0.03^2), shuffled, each
    with a random proper rotation and a translation N(0, 5^2).  Intra-cluster
    RMSD ~0.07 A, inter-cluster ~1.5 A: no pair near the 0.5 A threshold.
    Returns (coords (N, A, 3), atoms (A,) all 'C', cluster id per conformer)."""
    rng = np.random.default_rng(seed)
    skel = synthetic_skeleton(n_atoms, rng)
    K = max(1, n_conf // cluster_size)
    centres = np.empty((K, n_atoms, 3))
    for k in range(K):
        while True:
            c = skel + rng.normal(scale=sigma_cluster, size=skel.shape)
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
