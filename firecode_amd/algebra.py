"""Drop-in for the geometry names of ``firecode.algebra`` and
``prism_pruner.algebra`` that sit on the hot path."""

import numpy as np

from firecode_amd import _lib as L
from firecode_amd.rmsd import get_alignment_matrices


def align_vec_pair_batch(ref, tgt):
    """Batched ``align_vec_pair``: ref, tgt (K, 2, 3) -> (K, 3, 3)."""
    return get_alignment_matrices(ref, tgt)


def align_vec_pair(ref, tgt):
    """firecode/algebra.py:28-49: rotation that, applied to tgt, optimally
    aligns it to ref (two 3-vectors each)."""
    return np.ascontiguousarray(align_vec_pair_batch(L.f64(ref)[None], L.f64(tgt)[None])[0])


def count_clashes_batch(structures, lo=0.0, hi=0.5):
    """``count_clashes`` for every structure of an (N, A, 3) block."""
    X = L.f64(structures)
    if X.ndim != 3 or X.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {X.shape}")
    out = np.zeros(X.shape[0], dtype=np.int64)
    L.call("fc_clash_self", L.pf(X), X.shape[0], X.shape[1], float(lo), float(hi), L.pi(out))
    return out


def count_clashes(coords):
    """firecode/algebra.py:52-54: number of (ordered) atom pairs with 0 < d < 0.5."""
    return int(count_clashes_batch(L.f64(coords)[None])[0])


def get_inertia_moments_batch(structures, masses):
    X = L.f64(structures)
    masses = L.f64(masses)
    if X.ndim != 3 or X.shape[2] != 3 or masses.shape != (X.shape[1],):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "structures must be (N, A, 3) and masses (A,)")
    out = np.empty((X.shape[0], 3))
    L.call("fc_inertia_moments", L.pf(X), X.shape[0], X.shape[1], L.pf(masses), L.pf(out))
    return out


def get_inertia_moments(coords, masses):
    """prism_pruner.algebra.get_inertia_moments (call sites
    firecode/hypermolecule_class.py:66,72): three principal moments, ascending."""
    return get_inertia_moments_batch(L.f64(coords)[None], masses)[0]


def dihedral(p):
    """prism_pruner.algebra.dihedral([p1, p2, p3, p4]) -> degrees in (-180, 180]."""
    X = L.f64(p).reshape(1, 4, 3)
    out = np.empty((1, 1))
    quads = np.array([[0, 1, 2, 3]], dtype=np.int64)
    L.call("fc_torsion_fingerprint", L.pf(X), 1, 4, L.pi(quads), 1, L.pf(out))
    return float(out[0, 0])
