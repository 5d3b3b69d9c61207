"""Drop-in for ``prism_pruner.pruner`` (imported at firecode/ensemble.py:31,
embedder.py:45, operators.py:33): similarity pruning on the GPU.

Every function returns ``(structures[mask], mask)`` with ``mask`` a NumPy
bool array in the caller's order, like the reference call sites expect
(ensemble.py:211-235, embedder.py:1452-1496)."""

from time import perf_counter

import numpy as np

from firecode_amd import _lib as L
from firecode_amd.pt import pt


def _sorted_by_energy(structures, energies):
    """The reference processes structures in ascending-energy order when
    energies are given; same ``np.argsort`` call."""
    if energies is None:
        return None, None
    energies = np.asarray(energies, dtype=np.float64)
    if energies.shape[0] != structures.shape[0] or structures.shape[0] == 0:
        return None, None
    order = np.argsort(energies)
    return order, np.ascontiguousarray(energies[order])


def _unsort(mask_sorted, order):
    if order is None:
        return mask_sorted
    mask = np.empty_like(mask_sorted)
    mask[order] = mask_sorted
    return mask


def prune_by_rmsd(structures, atoms, max_rmsd=0.25, max_dev=None, energies=None, max_dE=0.0,
                  debugfunction=None, heavy_atoms_only=True, min_per_group=20):
    """Heavy-atom Kabsch-RMSD pruning: a pair is similar when
    ``rmsd < max_rmsd and maxdev < max_dev`` (default ``2*max_rmsd``)."""
    t0 = perf_counter()
    structures = L.f64(structures)
    if structures.ndim != 3 or structures.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
    atoms = np.asarray(atoms)
    if atoms.shape[0] != structures.shape[1]:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "len(atoms) != number of atoms")
    if max_dev is None:
        max_dev = 2 * max_rmsd
    N = structures.shape[0]
    if N == 0:
        return structures, np.ones(0, dtype=bool)
    heavy = (atoms != "H") if heavy_atoms_only else np.ones(len(atoms), dtype=bool)
    order, en_sorted = _sorted_by_energy(structures, energies)
    X = structures if order is None else np.ascontiguousarray(structures[order])
    with L.DeviceEnsemble(X, atom_mask=heavy, center=True) as ens:
        mask_sorted, stats = ens.prune(max_rmsd, max_dev, energies=en_sorted, max_dE=max_dE,
                                       min_per_group=min_per_group)
    mask = _unsort(mask_sorted, order)
    if debugfunction is not None:
        debugfunction(
            f"DEBUG: prune_by_rmsd [gfx950] - {stats[0]} pairs screened, {stats[1]} refined, "
            f"{stats[2]} similar, {stats[3]} grey, {stats[4]} ladder levels, "
            f"keeping {int(mask.sum())}/{N} in {perf_counter() - t0:.3f} s")
    return structures[mask], mask


def prune_by_moment_of_inertia(structures, atoms, max_deviation=0.01, energies=None, max_dE=0.0,
                               debugfunction=None, min_per_group=20):
    """MOI pruning: similar when all three principal moments differ by less
    than ``max_deviation`` relative to the earlier structure of the pair."""
    t0 = perf_counter()
    structures = L.f64(structures)
    if structures.ndim != 3 or structures.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
    N, A = structures.shape[0], structures.shape[1]
    if N == 0:
        return structures, np.ones(0, dtype=bool)
    masses = np.array([pt.mass(a) for a in atoms], dtype=np.float64)
    order, en_sorted = _sorted_by_energy(structures, energies)
    X = structures if order is None else np.ascontiguousarray(structures[order])
    mask8 = np.zeros(N, dtype=np.uint8)
    L.call("fc_prune_moi", L.pf(X), N, A, L.pf(masses), float(max_deviation), L.pf(en_sorted),
           float(max_dE), int(min_per_group), L.pb(mask8))
    mask = _unsort(mask8.astype(bool), order)
    if debugfunction is not None:
        debugfunction(f"DEBUG: prune_by_moment_of_inertia [gfx950] - keeping {int(mask.sum())}/{N} "
                      f"in {perf_counter() - t0:.3f} s")
    return structures[mask], mask


def prune(structures, atoms, max_rmsd=0.25, energies=None, max_dE=0.0, logfunction=None,
          debugfunction=None):
    """Combined pipeline used by firecode/interfaces/goat.py:399: MOI, then RMSD.
    (The symmetry-corrected stage is a later row of SURVEY.md section 8f.)"""
    structures = L.f64(structures)
    n0 = len(structures)
    s1, m1 = prune_by_moment_of_inertia(structures, atoms, energies=energies, max_dE=max_dE,
                                        debugfunction=debugfunction)
    e1 = None if energies is None else np.asarray(energies)[m1]
    s2, m2 = prune_by_rmsd(s1, atoms, max_rmsd, energies=e1, max_dE=max_dE, debugfunction=debugfunction)
    mask = np.zeros(n0, dtype=bool)
    mask[np.flatnonzero(m1)[m2]] = True
    if logfunction is not None:
        logfunction(f"Discarded {n0 - int(mask.sum())} candidates for MOI+RMSD similarity ({int(mask.sum())} left)")
    return structures[mask], mask


def greedy_prune_from_bits(bits, n, min_per_group=20):
    """k-ladder replay over a caller-supplied (n, ceil(n/64)) uint64 bit matrix."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    mask = np.zeros(n, dtype=np.uint8)
    L.call("fc_greedy_prune_from_bits", L.pw(bits), int(n), int(min_per_group), L.pb(mask))
    return mask.astype(bool)
