"""Drop-in for ``prism_pruner.rmsd`` (imported at firecode/utils.py:45,
embedder.py:46, hypermolecule_class.py:27): Kabsch superposition on the GPU."""

import numpy as np

from firecode_amd import _lib as L


def rmsd_and_max_batch(structures, pair_i, pair_j, center=False, atom_mask=None):
    """(rmsd, maxdev) of P conformer pairs of one (N, A, 3) block -- the batched
    form the one-pair name wraps (fc_kabsch_rmsd_pairs)."""
    X = L.f64(structures)
    if X.ndim != 3 or X.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {X.shape}")
    pi_, pj_ = L.i64(pair_i), L.i64(pair_j)
    P = int(pi_.shape[0])
    r, m = np.empty(P), np.empty(P)
    am = None if atom_mask is None else L.u8(np.asarray(atom_mask, dtype=bool))
    L.call("fc_kabsch_rmsd_pairs", L.pf(X), X.shape[0], X.shape[1], L.pb(am), L.pi(pi_), L.pi(pj_), P,
           int(bool(center)), L.pf(r), L.pf(m))
    return r, m


def rmsd_and_max(p, q, center=False):
    """``rmsd_and_max(p, q, center=False) -> (rmsd, maxdev)`` as called at
    firecode/utils.py:499, embedder.py:1784, ase_manipulations.py:1384."""
    X = np.stack([L.f64(p), L.f64(q)])
    r, m = rmsd_and_max_batch(X, [0], [1], center=center)
    return float(r[0]), float(m[0])


def get_alignment_matrices(P, Q):
    """Batched ``get_alignment_matrix``: P, Q (K, A, 3) -> (K, 3, 3)."""
    P, Q = L.f64(P), L.f64(Q)
    if P.shape != Q.shape or P.ndim != 3 or P.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"P, Q must both be (K, A, 3): {P.shape} vs {Q.shape}")
    M = np.empty((P.shape[0], 3, 3))
    L.call("fc_alignment_matrices", L.pf(P), L.pf(Q), P.shape[0], P.shape[1], L.pf(M))
    return M


def get_alignment_matrix(p, q):
    """Rotation M with ``(M @ q.T).T`` best superposed on p
    (firecode/hypermolecule_class.py:77-84).  Never raises LinAlgError:
    the quaternion solve always returns a proper rotation."""
    return get_alignment_matrices(L.f64(p)[None], L.f64(q)[None])[0]
