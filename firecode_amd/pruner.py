"""Drop-in for ``prism_pruner.pruner`` (imported at firecode/ensemble.py:31,
embedder.py:45, operators.py:33): similarity pruning on the GPU.

Every function returns ``(structures[mask], mask)`` with ``mask`` a NumPy
bool array in the caller's order, like the reference call sites expect
(ensemble.py:211-235, embedder.py:1452-1496)."""

from time import perf_counter

import numpy as np

from firecode_amd import _lib as L
from firecode_amd.pt import pt


# The conventions of prism_pruner's pruners that the reference tree does not show (SURVEY.md
# Appendix A: every call site treats the package as a black box), one named switch each, same names
# and defaults as the CONVENTIONS table of the test oracle.  Once `tests/golden/make_golden_prism.py` has been run
# where the package is installed (tests/test_prism_golden.py then says which values reproduce its masks)
# a differing convention is a one-line change here:
#   strict_lt        similar <=> rmsd < thr and maxdev < max_dev (True), or <= (False)
#   maxdev_factor    max_dev = maxdev_factor * max_rmsd when not given (firecode/utils.py:501: 2)
#   drop             "earlier": a structure is removed at the first later similar one; "later": mirror rule
#   default_max_rmsd threshold of a call that passes none (firecode/ensemble.py:230-235)
#   window_strict    pairs are comparable iff |dE| < max_dE (True) or <= (False)
#   moi_tolerance    relative tolerance of prune_by_moment_of_inertia (CHANGELOG.md:256: 1 %)
CONVENTIONS = {"strict_lt": True, "maxdev_factor": 2.0, "drop": "earlier", "default_max_rmsd": 0.25,
               "window_strict": True, "moi_tolerance": 0.01}


def _thresholds(max_rmsd, max_dev, max_dE):
    """The conventions as the kernels see them: `<=` is `<` against the next double up."""
    cv = CONVENTIONS
    max_rmsd = cv["default_max_rmsd"] if max_rmsd is None else float(max_rmsd)
    max_dev = cv["maxdev_factor"] * max_rmsd if max_dev is None else float(max_dev)
    if cv["drop"] not in ("earlier", "later"):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"CONVENTIONS['drop'] = {cv['drop']!r}")
    L.call("fc_prune_conventions", int(cv["drop"] == "later"))
    if not cv["strict_lt"]:
        max_rmsd, max_dev = np.nextafter(max_rmsd, np.inf), np.nextafter(max_dev, np.inf)
    if not cv["window_strict"]:
        max_dE = np.nextafter(float(max_dE), np.inf)
    return float(max_rmsd), float(max_dev), float(max_dE)


def _sorted_by_energy(structures, energies):
    """The reference processes structures in ascending-energy order when
    energies are given; same ``np.argsort`` call."""
    if energies is None:
        return None, None
    energies = np.asarray(energies, dtype=np.float64)
    if energies.shape[0] != structures.shape[0] or structures.shape[0] == 0:
        return None, None
    # stable: a later stage that sorts the SURVIVORS of an earlier one then processes them in the order
    # the earlier stage left them in -- what makes the fused pipeline equal to the stage-by-stage calls
    order = np.argsort(energies, kind="stable")
    return order, np.ascontiguousarray(energies[order])


def _unsort(mask_sorted, order):
    if order is None:
        return mask_sorted
    mask = np.empty_like(mask_sorted)
    mask[order] = mask_sorted
    return mask


def prune_by_rmsd(structures, atoms, max_rmsd=None, max_dev=None, energies=None, max_dE=0.0,
                  debugfunction=None, heavy_atoms_only=True, min_per_group=20):
    """Heavy-atom Kabsch-RMSD pruning: a pair is similar when
    ``rmsd < max_rmsd and maxdev < max_dev`` (default ``2*max_rmsd``; ``max_rmsd`` defaults to
    ``CONVENTIONS["default_max_rmsd"]``, the case of firecode/ensemble.py:230 which passes none)."""
    t0 = perf_counter()
    structures = L.f64(structures)
    if structures.ndim != 3 or structures.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
    atoms = np.asarray(atoms)
    if atoms.shape[0] != structures.shape[1]:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "len(atoms) != number of atoms")
    max_rmsd, max_dev, max_dE = _thresholds(max_rmsd, max_dev, max_dE)
    N = structures.shape[0]
    if N == 0:
        return structures, np.ones(0, dtype=bool)
    heavy = (atoms != "H") if heavy_atoms_only else np.ones(len(atoms), dtype=bool)
    order, en_sorted = _sorted_by_energy(structures, energies)
    X = structures if order is None else np.ascontiguousarray(structures[order])
    # one C call (fc_prune_rmsd_host): upload, preparation, prune, mask
    m8 = np.zeros(N, dtype=np.uint8)
    stats = np.zeros(6, dtype=np.int64)
    hm = np.ascontiguousarray(heavy, dtype=np.uint8)
    L.call("fc_prune_rmsd_host", L.pf(X), N, X.shape[1], L.pb(hm), 1, float(max_rmsd), float(max_dev),
           L.pf(None if en_sorted is None else L.f64(en_sorted)), float(max_dE), int(min_per_group), L.pb(m8), L.pi(stats))
    mask = _unsort(m8.view(np.bool_), order)
    if debugfunction is not None:
        debugfunction(
            f"DEBUG: prune_by_rmsd [gfx950] - {stats[0]} pairs screened, {stats[1]} refined, "
            f"{stats[2]} similar, {stats[3]} grey, {stats[4]} ladder levels, "
            f"keeping {int(mask.sum())}/{N} in {perf_counter() - t0:.3f} s")
    return structures[mask], mask


def rotation_mask(graph, torsion, n_atoms=None):
    """``_get_rotation_mask`` (firecode/torsion_module.py:354-382): the atoms that rotate
    with i4 -- reachable from i4 once the i2-i3 edge is removed -- with i3 excluded."""
    import networkx as nx

    _, i2, i3, i4 = (int(t) for t in torsion)
    n = graph.number_of_nodes() if n_atoms is None else n_atoms
    had = graph.has_edge(i2, i3)
    if had:
        graph.remove_edge(i2, i3)
    try:
        reach = nx.node_connected_component(graph, i4)
    finally:
        if had:
            graph.add_edge(i2, i3)
    mask = np.zeros(n, dtype=bool)
    mask[list(reach)] = True
    mask[i3] = False
    return mask


def prune_many_by_rmsd(ensembles, max_rmsd=None, max_dev=None, heavy_atoms_only=True, min_per_group=20):
    """``prune_by_rmsd`` for a queue of ensembles: ``ensembles`` is a list of
    ``(structures, atoms)``; returns a list of ``(structures[mask], mask)``, each identical to
    what ``prune_by_rmsd(structures, atoms, max_rmsd, max_dev)`` returns.  All ensembles are made
    resident, their prunes are enqueued together (``fc_prune_rmsd_many``) and the host waits once;
    the reference prunes one ensemble per call (the loop a maintainer would write around
    firecode/ensemble.py:230-235)."""
    max_rmsd, max_dev, _ = _thresholds(max_rmsd, max_dev, 0.0)
    items, resident = [], []
    try:
        for structures, atoms in ensembles:
            structures = L.f64(structures)
            if structures.ndim != 3 or structures.shape[2] != 3:
                raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
            atoms = np.asarray(atoms)
            if atoms.shape[0] != structures.shape[1]:
                raise L.FirecodeHipInputError(L.FC_E_INVALID, "len(atoms) != number of atoms")
            items.append(structures)
            if structures.shape[0] == 0:
                resident.append(None)
                continue
            heavy = (atoms != "H") if heavy_atoms_only else np.ones(len(atoms), dtype=bool)
            resident.append(L.DeviceEnsemble(structures, atom_mask=heavy, center=True))
        live = [e for e in resident if e is not None]
        masks, _ = L.prune_many(live, max_rmsd, max_dev, min_per_group) if live else ([], None)
    finally:
        for e in resident:
            if e is not None:
                e.close()
    out, it = [], iter(masks)
    for structures, e in zip(items, resident):
        mask = np.ones(0, dtype=bool) if e is None else next(it)
        out.append((structures[mask], mask))
    return out


def prune_by_rmsd_rot_corr(structures, atoms, graph=None, max_rmsd=0.25, max_dev=None, energies=None, max_dE=0.0,
                           logfunction=None, debugfunction=None, torsions=None, rotation_masks=None,
                           min_per_group=20, return_bits=False):
    """``prune_by_rmsd_rot_corr`` (prism_pruner.pruner; call sites firecode/ensemble.py:253-260,
    embedder.py:1489-1496, operators.py:626-632): RMSD pruning that is invariant to rotations of
    locally symmetric groups (tBu, Ph, NMe2 ...).

    Called as the reference calls it -- ``(structures, atoms, graph, max_rmsd=..., energies=...,
    max_dE=..., logfunction=..., debugfunction=...)`` -- the locally symmetric torsions are
    perceived from ``graph`` (``firecode_amd.torsion_perception.symmetric_torsions``: the in-tree
    ``_get_torsions(..., keepdummy=True, mode="symmetry")`` filtered to dummy rotations).
    ``torsions=`` may instead give them explicitly as ``(i1, i2, i3, i4, n_fold)`` (an empty list:
    the molecule has none, plain heavy-atom RMSD prune); without both ``graph`` and ``torsions``
    the call is refused -- it never runs as an uncorrected prune under this name.
    ``rotation_masks`` (T, A) may be given, or are derived from ``graph`` like
    ``_get_rotation_mask`` does.  PARITY UNPINNED (third-party algorithm restated, see
    include/fc_hip.h)."""
    from firecode_amd.torsion_module import N_FOLD_ANGLES

    t0 = perf_counter()
    structures = L.f64(structures)
    if structures.ndim != 3 or structures.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
    atoms = np.asarray(atoms)
    N, A = structures.shape[:2]
    if atoms.shape[0] != A:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "len(atoms) != number of atoms")
    max_rmsd, max_dev, max_dE = _thresholds(max_rmsd, max_dev, max_dE)
    if N == 0:
        return structures, np.ones(0, dtype=bool)
    if torsions is None:
        if graph is None:
            raise L.FirecodeHipInputError(
                L.FC_E_INVALID, "prune_by_rmsd_rot_corr needs the molecular graph (third positional argument, as "
                "firecode/ensemble.py:253 passes it) or torsions=: without them no symmetry correction is possible")
        from firecode_amd.torsion_perception import symmetric_torsions

        torsions = symmetric_torsions(graph, structures[0], atoms)
    torsions = list(torsions)
    T = len(torsions)
    quads = L.i64(np.array([t[:4] for t in torsions], dtype=np.int64).reshape(T, 4))
    if rotation_masks is None:
        if T and graph is None:
            raise L.FirecodeHipInputError(L.FC_E_INVALID, "rotation_masks or graph is required with torsions")
        rotation_masks = [rotation_mask(graph, t[:4], A) for t in torsions]
    masks = L.u8(np.asarray(rotation_masks, dtype=bool).reshape(T, A))
    sets = [N_FOLD_ANGLES[int(t[4])] for t in torsions]
    max_angles = max([len(a) for a in sets] + [1])
    angles = np.zeros((T, max_angles))
    n_angles = np.zeros(T, dtype=np.int32)
    for k, a in enumerate(sets):
        angles[k, : len(a)] = a
        n_angles[k] = len(a)
    heavy = L.u8(atoms != "H")
    order, en_sorted = _sorted_by_energy(structures, energies)
    X = structures if order is None else np.ascontiguousarray(structures[order])
    mask_sorted = np.zeros(N, dtype=np.uint8)
    W = (N + 63) // 64
    bits = np.zeros((N, W), dtype=np.uint64) if return_bits else None
    import ctypes as C

    L.call("fc_prune_rmsd_rot_corr", L.pf(X), N, A, L.pb(heavy), L.pi(quads), T, L.pb(masks), L.pf(angles),
           n_angles.ctypes.data_as(C.POINTER(C.c_int32)), max_angles, float(max_rmsd), float(max_dev),
           None if en_sorted is None else L.pf(L.f64(en_sorted)), float(max_dE), int(min_per_group),
           L.pb(mask_sorted), None if bits is None else L.pw(bits))
    mask = _unsort(mask_sorted.astype(bool), order)
    for fn in (logfunction, debugfunction):
        if fn is not None:
            fn(f"DEBUG: prune_by_rmsd_rot_corr [gfx950] - {T} symmetric torsions, keeping {int(mask.sum())}/{N} "
               f"in {perf_counter() - t0:.3f} s")
    if return_bits:
        return structures[mask], mask, bits
    return structures[mask], mask


def prune_by_moment_of_inertia(structures, atoms, max_deviation=None, energies=None, max_dE=0.0,
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
    if max_deviation is None:
        max_deviation = CONVENTIONS["moi_tolerance"]
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


def prune_similarity(structures, atoms, moi=True, rmsd=True, max_rmsd=None, max_dev=None, max_deviation=None,
                     energies=None, max_dE=0.0, heavy_atoms_only=True, min_per_group=20):
    """The MOI and RMSD stages of ``Ensemble.similarity_pruning`` (firecode/ensemble.py:205-235) /
    ``Embedder.similarity_refining`` (embedder.py:1445-1474) on ONE upload of the coordinates
    (``fc_prune_similarity``): the MOI stage runs on the resident structures, its survivors are gathered
    on the device into the RMSD stage's layout, the masks are composed on the way out.  Stage by
    stage the masks equal ``prune_by_moment_of_inertia`` followed by ``prune_by_rmsd`` on its output.
    Returns ``(mask_after_moi, mask_after_both, (n_in, n_after_moi, n_after_rmsd))`` in the caller's order."""
    structures = L.f64(structures)
    if structures.ndim != 3 or structures.shape[2] != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, f"structures must be (N, A, 3), got {structures.shape}")
    atoms = np.asarray(atoms)
    N, A = structures.shape[:2]
    if atoms.shape[0] != A:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "len(atoms) != number of atoms")
    max_rmsd, max_dev, max_dE = _thresholds(max_rmsd, max_dev, max_dE)
    if max_deviation is None:
        max_deviation = CONVENTIONS["moi_tolerance"]
    counts = np.array([N, N, N], dtype=np.int64)
    if N == 0:
        return np.ones(0, dtype=bool), np.ones(0, dtype=bool), counts
    heavy = L.u8((atoms != "H") if heavy_atoms_only else np.ones(A, dtype=bool))
    masses = np.array([pt.mass(a) for a in atoms], dtype=np.float64)
    order, en_sorted = _sorted_by_energy(structures, energies)
    X = structures if order is None else np.ascontiguousarray(structures[order])
    m1, m2 = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
    L.call("fc_prune_similarity", L.pf(X), N, A, L.pb(heavy), L.pf(masses), int(bool(moi)), float(max_deviation),
           int(bool(rmsd)), max_rmsd, max_dev, L.pf(en_sorted), max_dE, int(min_per_group), L.pb(m1), L.pb(m2),
           L.pi(counts))
    return _unsort(m1.astype(bool), order), _unsort(m2.astype(bool), order), counts


def prune(structures, atoms, max_rmsd=0.25, energies=None, max_dE=0.0, logfunction=None,
          debugfunction=None):
    """Combined pipeline used by firecode/interfaces/goat.py:399: MOI, then RMSD.
    (The symmetry-corrected stage is a later row of SURVEY.md section 8f.)"""
    structures = L.f64(structures)
    n0 = len(structures)
    _, mask, _ = prune_similarity(structures, atoms, max_rmsd=max_rmsd, energies=energies, max_dE=max_dE)
    for fn in (logfunction, debugfunction):
        if fn is not None:
            fn(f"Discarded {n0 - int(mask.sum())} candidates for MOI+RMSD similarity ({int(mask.sum())} left)")
    return structures[mask], mask


def greedy_prune_from_bits(bits, n, min_per_group=20):
    """k-ladder replay over a caller-supplied (n, ceil(n/64)) uint64 bit matrix."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    mask = np.zeros(n, dtype=np.uint8)
    L.call("fc_greedy_prune_from_bits", L.pw(bits), int(n), int(min_per_group), L.pb(mask))
    return mask.astype(bool)
