"""Drop-in for the numeric name of ``firecode.hypermolecule_class``."""

import numpy as np

from firecode_amd import _lib as L
from firecode_amd.pt import pt


def align_by_moi(atoms, structures, masses=None):
    """firecode/hypermolecule_class.py:45-86: align a (n_structures, n_atoms, 3) array
    to its first structure "based on the moments of inertia vectors": every structure
    is centred on its mean (the reference does that IN PLACE on the caller's array; so
    does this), then rotated by ``get_alignment_matrix(diag(I_ref), diag(I_target))``.
    ``masses`` overrides the periodic-table lookup of ``atoms``.  Never raises
    LinAlgError (the reference's fallback for that case is the identity; the quaternion
    solve used here always converges)."""
    X = L.f64(structures)
    if X.ndim != 3 or X.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {X.shape}")
    m = L.f64([pt.mass(el) for el in atoms] if masses is None else masses)
    if m.shape != (X.shape[1],):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "one mass per atom is required")
    out = np.empty_like(X)
    L.call("fc_align_by_moi", L.pf(X), X.shape[0], X.shape[1], L.pf(m), L.pf(out))
    if isinstance(structures, np.ndarray) and structures.dtype == np.float64 and len(structures):
        structures -= structures.mean(axis=1, keepdims=True)  # the reference's in-place centring
    return out
