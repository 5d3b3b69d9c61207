"""Drop-in for the numeric core of ``firecode.embeds``."""

import numpy as np

from firecode_amd import _lib as L


def rototranslate(coords, R, t):
    """(n, A, 3) blocks, R (n, 3, 3), t (n, 3) -> ``(R @ X.T).T + t`` per block."""
    X, R, t = L.f64(coords), L.f64(R), L.f64(t)
    if X.ndim != 3 or R.shape != (X.shape[0], 3, 3) or t.shape != (X.shape[0], 3):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "coords (n, A, 3), R (n, 3, 3), t (n, 3) expected")
    out = np.empty_like(X)
    L.call("fc_rototranslate", L.pf(X), X.shape[0], X.shape[1], L.pf(R), L.pf(t), L.pf(out))
    return out


def get_embed(mols, conf_ids):
    """firecode/embeds.py:808-817: concatenated roto-translated molecules; mols
    carry ``.coords (n_conf, A, 3)``, ``.rotation (3, 3)``, ``.position (3,)``."""
    return np.concatenate([
        rototranslate(L.f64(mol.coords[c])[None], L.f64(mol.rotation)[None], L.f64(mol.position)[None])[0]
        for mol, c in zip(mols, conf_ids)
    ])


def embed_poses_clash(m1, m2, c1, c2, R1, t1, R2, t2, thresh=1.5, max_clashes=0, return_poses=False):
    """Clash test of the bimolecular rigid-embed loop (embeds.py:713-722) for P
    poses at once: pose k = conformer c1[k] of m1 under (R1[k], t1[k]) next to
    conformer c2[k] of m2 under (R2[k], t2[k]).
    Returns (pass (P,) bool, counts (P,) int64[, poses (P, A1+A2, 3)])."""
    m1, m2 = L.f64(m1), L.f64(m2)
    c1, c2 = L.i64(c1), L.i64(c2)
    R1, t1, R2, t2 = L.f64(R1), L.f64(t1), L.f64(R2), L.f64(t2)
    P = c1.shape[0]
    if not (c2.shape == (P,) and R1.shape == (P, 3, 3) and R2.shape == (P, 3, 3)
            and t1.shape == (P, 3) and t2.shape == (P, 3) and m1.ndim == 3 and m2.ndim == 3):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "inconsistent pose arrays")
    ok = np.zeros(P, dtype=np.uint8)
    counts = np.zeros(P, dtype=np.int64)
    poses = np.empty((P, m1.shape[1] + m2.shape[1], 3)) if return_poses else None
    L.call("fc_embed_poses_clash", L.pf(m1), m1.shape[0], m1.shape[1], L.pf(m2), m2.shape[0], m2.shape[1],
           L.pi(c1), L.pi(c2), L.pf(R1), L.pf(t1), L.pf(R2), L.pf(t2), P, float(thresh), int(max_clashes),
           L.pi(counts), L.pb(ok), L.pf(poses))
    if return_poses:
        return ok.astype(bool), counts, poses
    return ok.astype(bool), counts
