"""Array-level mirrors of the pruning stages of ``Embedder``
(firecode/embedder.py:1352-1514, 1954-2039): each function takes the arrays
the method reads from ``self`` and returns the boolean mask the method applies
with ``apply_mask`` -- so ``RunEmbedding.run`` can keep its bookkeeping and call
these for the arithmetic."""

from time import perf_counter

import numpy as np

from firecode_amd.pruner import prune_by_moment_of_inertia, prune_by_rmsd
from firecode_amd.utils import compenetration_check_batch, fitness_check_batch


def _log(logfunction, msg=""):
    if logfunction is not None:
        logfunction(msg)


def compenetration_refining(structures, ids=None, graph=None, clash_thresh=1.5, max_clashes=0, logfunction=None):
    """embedder.py:1954-1995: per-structure ``compenetration_check`` -> mask."""
    t0 = perf_counter()
    _log(logfunction, "--> Checking structures for compenetrations")
    mask = compenetration_check_batch(structures, graph=graph, ids=ids, thresh=clash_thresh,
                                      max_clashes=max_clashes)
    if not mask.all():
        _log(logfunction, f"Discarded {int((~mask).sum())} candidates for compenetration "
                          f"({int(mask.sum())} left, {perf_counter() - t0:.3f} s)")
    else:
        _log(logfunction, f"All {len(mask)} structures passed the compenetration check")
    return mask


def fitness_refining(structures, constrained_indices, constrained_distances, threshold=5.0, logfunction=None):
    """embedder.py:1997-2039: keep structures whose summed deviation from the
    target pairing distances is below ``threshold``."""
    mask, _ = fitness_check_batch(structures, constrained_indices, constrained_distances, threshold)
    if not mask.all():
        _log(logfunction, f"Discarded {int((~mask).sum())} candidates for unfitness ({int(mask.sum())} left)")
    return mask


def first_threshold_keeping(rel, kcal_thresh, keep_min):
    """The rule shared by ensemble.py:134-169 and embedder.py:1365-1395 without their loop: ``kcal_thresh`` when
    it keeps more than ``keep_min`` of the structures, else the first relative energy (in array order) above
    it that does, else ``kcal_thresh`` -> (threshold, fraction kept by it or None when unchanged).  The number of
    structures strictly below a candidate is its rank in the sorted energies: one sort, one searchsorted."""
    rel = np.asarray(rel, dtype=np.float64)
    n = len(rel)
    if np.count_nonzero(rel < kcal_thresh) / n > keep_min:
        return kcal_thresh, None
    below = np.searchsorted(np.sort(rel), rel, side="left")
    ok = (rel > kcal_thresh) & (below / n > keep_min)
    if not ok.any():
        return kcal_thresh, None
    first = int(np.argmax(ok))
    return float(rel[first]), below[first] / n


def dynamic_energy_thr(energies, kcal_thresh=10.0, keep_min=0.1, logfunction=None):
    """embedder.py:1365-1395."""
    thr, kept = first_threshold_keeping(np.asarray(energies) - np.min(energies), kcal_thresh, keep_min)
    if kept is not None:
        _log(logfunction, f"--> Dynamically adjusted energy threshold to {thr:.1f} kcal/mol to retain at "
                          f"least {kept * 100:.2f}% of structures.")
    return thr


def energy_pruning(energies, kcal_thresh=10.0, logfunction=None):
    """embedder.py:1352-1363 -> mask ``rel_energies < dynamic threshold``."""
    energies = np.asarray(energies, dtype=np.float64)
    thr = dynamic_energy_thr(energies, kcal_thresh, logfunction=logfunction)
    mask = (energies - energies.min()) < thr
    if not mask.all():
        _log(logfunction, f"Discarded {int((~mask).sum())} candidates for energy ({int(mask.sum())} left, "
                          f"{round(100 * mask.sum() / len(mask), 1)}% kept, threshold {thr:.1f} kcal/mol)")
    return mask


def similarity_refining(structures, atoms, rmsd_thr=0.5, quadruplets=None, tfd=False, moi=True, rmsd=True,
                        max_structures=None, logfunction=None, debugfunction=None, rmsd_rot_corr=False,
                        symmetric_torsions=None, graph=None, rotation_masks=None):
    """embedder.py:1410-1514: [TFD] -> MOI -> RMSD -> [symmetry-corrected RMSD, at most 1000
    structures and only with the molecular graph (``embed_graph``), :1480-1505] (no energies are passed
    at these call sites).  ``max_structures``: the reference skips MOI/RMSD above
    1e5 structures (embedder.py:1446,1467); the GPU path has no such cap unless
    one is given.  Returns the cumulative mask over the input structures."""
    structures = np.asarray(structures, dtype=np.float64)
    alive = np.arange(len(structures))

    def stage(fn, label, *args, **kw):
        nonlocal alive
        t0 = perf_counter()
        _, m = fn(structures[alive], *args, **kw)
        if not m.all():
            _log(logfunction, f"Discarded {int((~m).sum())} candidates for {label} similarity "
                              f"({int(m.sum())} left, {perf_counter() - t0:.3f} s)")
        alive = alive[m]

    if tfd and quadruplets is not None and len(quadruplets) > 0:
        from firecode_amd.torsion_module import prune_conformers_tfd

        stage(prune_conformers_tfd, "TFD", quadruplets)
    capped = max_structures is not None and len(alive) > max_structures
    if moi and rmsd and not capped:
        # both stages on ONE upload (fc_prune_similarity): the MOI survivors are gathered on the device
        from firecode_amd.pruner import prune_similarity

        t0 = perf_counter()
        _, m_both, counts = prune_similarity(structures[alive], atoms, max_rmsd=rmsd_thr)
        dt = perf_counter() - t0
        for label, a, b in (("MOI", counts[0], counts[1]), ("RMSD", counts[1], counts[2])):
            if b < a:
                _log(logfunction, f"Discarded {int(a - b)} candidates for {label} similarity ({int(b)} left, {dt:.3f} s)")
        if debugfunction is not None:
            debugfunction(f"DEBUG: prune_similarity [gfx950] - MOI + RMSD on one upload, {int(counts[0])} -> "
                          f"{int(counts[1])} -> {int(counts[2])} in {dt:.3f} s")
        alive = alive[m_both]
    else:
        for flag, fn, label, args in ((moi, prune_by_moment_of_inertia, "MOI", (atoms,)),
                                      (rmsd, prune_by_rmsd, "RMSD", (atoms, rmsd_thr))):
            if not flag:
                continue
            if max_structures is not None and len(alive) > max_structures:
                _log(logfunction, f"Skipped {label} pruning (>{max_structures} structures)")
                continue
            stage(fn, label, *args, debugfunction=debugfunction)
    if rmsd_rot_corr and (symmetric_torsions is not None or graph is not None):
        # embedder.py:1485-1496: runs whenever the embedder holds the graph; the torsions come from
        # it unless the caller names them
        if len(alive) <= 1e3:
            from firecode_amd.pruner import prune_by_rmsd_rot_corr

            stage(lambda X: prune_by_rmsd_rot_corr(X, atoms, graph, max_rmsd=rmsd_thr, torsions=symmetric_torsions,
                                                   rotation_masks=rotation_masks, debugfunction=debugfunction),
                  "symmetry-corrected RMSD")
        else:
            _log(logfunction, "Skipped rotationally-corrected RMSD pruning (>1k structures)")
    mask = np.zeros(len(structures), dtype=bool)
    mask[alive] = True
    if mask.all():
        _log(logfunction, f"All structures passed the similarity check.{' ' * 15}")
    _log(logfunction)
    return mask
