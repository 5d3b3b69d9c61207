"""Drop-in for the numeric names of ``firecode.torsion_module``."""

import numpy as np

from firecode_amd import _lib as L


def torsion_scan(base, torsions, masks, angles, thresh=1.5, backoff=5, out=None):
    """Inner loops of ``clustered_csearch`` (firecode/torsion_module.py:812-856)
    for one starting structure: every row of ``angles`` (S, T) is applied to
    ``base`` (A, 3) with the clash test and the 5-degree back-off.
    Returns (coords (S, A, 3), rotated_bonds (S,)).  ``out``: a C-contiguous float64 (S, A, 3) array (or view)
    to write the conformers into -- a caller that puts the starting structure in front of them saves the copy
    of ``np.concatenate``."""
    base = L.f64(base)
    tors = L.i64(torsions).reshape(-1, 4)
    msk = L.u8(np.asarray(masks, dtype=bool)).reshape(tors.shape[0], -1)
    ang = L.i64(angles).reshape(-1, tors.shape[0])
    A, T, S = base.shape[0], tors.shape[0], ang.shape[0]
    if base.ndim != 2 or base.shape[1] != 3 or msk.shape[1] != A:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "base must be (A, 3) and masks (T, A)")
    if out is None:
        out = np.empty((S, A, 3))
    elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (S, A, 3) and out.flags.c_contiguous):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "out must be a C-contiguous float64 array of shape (S, A, 3)")
    rot = np.zeros(S, dtype=np.int64)
    L.call("fc_torsion_scan", L.pf(base), A, L.pi(tors), T, L.pb(msk), L.pi(ang), S, float(thresh),
           int(backoff), L.pf(out), L.pi(rot))
    return out, rot


def torsion_scan_fingerprints(base, torsions, masks, angles, quadruplets, thresh=1.5, backoff=5, want_coords=False):
    """``torsion_scan`` plus the torsion fingerprint (degrees, (S, Q)) of every generated
    conformer, taken inside the kernel; the (S, A, 3) conformers themselves are only
    returned on request.  Returns (tf, rotated_bonds[, coords])."""
    base = L.f64(base)
    tors = L.i64(torsions).reshape(-1, 4)
    msk = L.u8(np.asarray(masks, dtype=bool)).reshape(tors.shape[0], -1)
    ang = L.i64(angles).reshape(-1, tors.shape[0])
    quads = L.i64(quadruplets).reshape(-1, 4)
    A, T, S, Q = base.shape[0], tors.shape[0], ang.shape[0], quads.shape[0]
    if base.ndim != 2 or base.shape[1] != 3 or msk.shape[1] != A or Q == 0:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "base (A, 3), masks (T, A) and at least one quadruplet expected")
    tf = np.empty((S, Q))
    rot = np.zeros(S, dtype=np.int64)
    out = np.empty((S, A, 3)) if want_coords else None
    L.call("fc_torsion_scan_fingerprints", L.pf(base), A, L.pi(tors), T, L.pb(msk), L.pi(ang), S, float(thresh),
           int(backoff), L.pi(quads), Q, L.pf(tf), L.pi(rot), None if out is None else L.pf(out))
    return (tf, rot, out) if want_coords else (tf, rot)


def torsion_scan_tfd(base, torsions, masks, angles, quadruplets, thresh=1.5, backoff=5, tfd_thresh=10):
    """``torsion_scan_fingerprints`` + ``prune_tfd_from_tf_mat`` on ``[base] + [conformers that rotated a
    bond]`` in one call, fingerprints resident on the device (fc_torsion_scan_tfd).
    Returns (rotated_bonds (S,), keep (S + 1,) bool: [0] = the starting structure)."""
    base = L.f64(base)
    tors = L.i64(torsions).reshape(-1, 4)
    msk = L.u8(np.asarray(masks, dtype=bool)).reshape(tors.shape[0], -1)
    ang = L.i64(angles).reshape(-1, tors.shape[0])
    quads = L.i64(quadruplets).reshape(-1, 4)
    A, T, S, Q = base.shape[0], tors.shape[0], ang.shape[0], quads.shape[0]
    if base.ndim != 2 or base.shape[1] != 3 or msk.shape[1] != A or Q == 0:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "base (A, 3), masks (T, A) and at least one quadruplet expected")
    rot = np.zeros(S, dtype=np.int64)
    keep = np.zeros(S + 1, dtype=np.uint8)
    L.call("fc_torsion_scan_tfd", L.pf(base), A, L.pi(tors), T, L.pb(msk), L.pi(ang), S, float(thresh), int(backoff),
           L.pi(quads), Q, float(tfd_thresh), L.pi(rot), L.pb(keep))
    return rot, keep.astype(bool)


def torsion_scan_tfd_grid(base, torsions, masks, values, quadruplets, thresh=1.5, backoff=5, tfd_thresh=10):
    """``torsion_scan_tfd`` over ``cartesian_product(*values)`` -- ``values`` = one sequence of angles (degrees) per
    torsion -- with the grid generated on the device in the reference's row order (fc_torsion_scan_tfd_grid; the
    loop of firecode/torsion_module.py:822).  Row s of the result is row s of that product
    (``firecode_amd.utils.cartesian_rows_at`` gives its angles).  Returns (rotated_bonds (S,), keep (S + 1,) bool)."""
    base = L.f64(base)
    tors = L.i64(torsions).reshape(-1, 4)
    msk = L.u8(np.asarray(masks, dtype=bool)).reshape(tors.shape[0], -1)
    quads = L.i64(quadruplets).reshape(-1, 4)
    flat = [L.i64(np.asarray(v).reshape(-1)) for v in values]
    A, T, Q = base.shape[0], tors.shape[0], quads.shape[0]
    if len(flat) != T:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "one list of angles per torsion expected")
    if base.ndim != 2 or base.shape[1] != 3 or msk.shape[1] != A or Q == 0:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "base (A, 3), masks (T, A) and at least one quadruplet expected")
    counts = np.array([len(v) for v in flat], dtype=np.int64)
    S = int(np.prod(counts, dtype=object))
    vals = np.ascontiguousarray(np.concatenate(flat)) if flat else np.zeros(0, dtype=np.int64)
    if len(vals) == 0:
        vals = np.zeros(1, dtype=np.int64)
    rot = np.zeros(S, dtype=np.int64)
    keep = np.zeros(S + 1, dtype=np.uint8)
    L.call("fc_torsion_scan_tfd_grid", L.pf(base), A, L.pi(tors), T, L.pb(msk), L.pi(vals), L.pi(counts), float(thresh),
           int(backoff), L.pi(quads), Q, float(tfd_thresh), L.pi(rot), L.pb(keep))
    return rot, keep.astype(bool)


def torsion_comp_check(coords, torsion, mask, thresh=1.5, max_clashes=0):
    """firecode/torsion_module.py:894-918: rest-vs-moving clash count <= max_clashes."""
    X = L.f64(coords)
    mask = np.asarray(mask, dtype=bool)
    _, i2, i3, _ = torsion
    anti = ~mask
    anti[i2] = False
    anti[i3] = False
    # fragment form: [moving | rest] contiguous, bimolecular strict-< count
    packed = np.concatenate([X[mask], X[anti]])[None]
    if packed.shape[1] == 0 or mask.sum() == 0 or anti.sum() == 0:
        return True
    ok = np.zeros(1, dtype=np.uint8)
    ids = np.array([int(mask.sum()), int(anti.sum())], dtype=np.int64)
    L.call("fc_clash_fragments", L.pf(L.f64(packed)), 1, packed.shape[1], L.pi(ids), 2, float(thresh),
           int(max_clashes), None, L.pb(ok))
    return bool(ok[0])


def get_tf_mat(structures, quadruplets):
    """firecode/torsion_module.py:1046-1053 (``_get_tf_mat``), batched."""
    X = L.f64(structures)
    quads = L.i64(quadruplets).reshape(-1, 4)
    out = np.zeros((X.shape[0], quads.shape[0]))
    L.call("fc_torsion_fingerprint", L.pf(X), X.shape[0], X.shape[1], L.pi(quads), quads.shape[0], L.pf(out))
    return out


def get_torsion_fingerprint(coords, quadruplets):
    """firecode/torsion_module.py:1070-1076."""
    return get_tf_mat(L.f64(coords)[None], quadruplets)[0]


def tfd_simbits(tf_mat, thresh=10, row_begin=0, row_end=None):
    tf = L.f64(tf_mat)
    N, Q = tf.shape
    row_end = N if row_end is None else row_end
    W = (N + 63) // 64
    bits = np.zeros((row_end - row_begin, W), dtype=np.uint64)
    L.call("fc_tfd_simbits", L.pf(tf), N, Q, float(thresh), row_begin, row_end, L.pw(bits))
    return bits


def tfd_similarity(tfp1, tfp2, thresh=10):
    """firecode/torsion_module.py:1056-1067."""
    tf = np.stack([L.f64(tfp1), L.f64(tfp2)])
    bits = tfd_simbits(tf, thresh, 0, 1)
    return bool((int(bits[0, 0]) >> 1) & 1)


def prune_tfd_from_tf_mat(tf_mat, thresh=10):
    """The k-ladder of ``prune_conformers_tfd`` given the fingerprint matrix:
    the GPU finds, for every structure, its first TFD-similar successor
    (O(N^2) comparisons, N integers out); the reference's per-chunk
    match-graph / "keep group[0]" bookkeeping is replayed from that array in
    C++ (csrc/fc_tfd_host.cpp), reproducing CPython's set order so that the
    mask equals the reference's bit for bit."""
    tf = L.f64(tf_mat)
    if tf.ndim != 2:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "tf_mat must be (N, Q)")
    mask = np.zeros(tf.shape[0], dtype=np.uint8)
    L.call("fc_tfd_prune", L.pf(tf), tf.shape[0], tf.shape[1], float(thresh), L.pb(mask))
    return mask.astype(bool)


def prune_conformers_tfd(structures, quadruplets, thresh=10, verbose=False):
    """firecode/torsion_module.py:957-1043 -> (structures[mask], mask)."""
    structures = L.f64(structures)
    mask = prune_tfd_from_tf_mat(get_tf_mat(structures, quadruplets), thresh)
    return structures[mask], mask


def most_diverse_conformers(n, structures, seed=None):
    """firecode/torsion_module.py:574-586: everything when there are at most n structures (:581-582),
    else a RANDOM subsample of n (with replacement, sorted indices, :585-586).  The reference draws
    from the global, unseeded NumPy RNG; pass ``seed`` for a reproducible draw."""
    if len(structures) <= n:
        return list(np.array(structures))
    rng = np.random if seed is None else np.random.RandomState(seed)
    indices = np.sort(rng.choice(len(structures), size=n))
    return list(np.array(structures)[indices])


def clustered_csearch_core(coords, torsions, rotation_masks, n_out=100, thresh=1.5, seed=None, logfunction=None):
    """Numeric core of ``clustered_csearch`` (torsion_module.py:726-891) for one
    torsion group: ``torsions`` = sequence of (i1, i2, i3, i4, n_fold),
    ``rotation_masks`` = the matching ``_get_rotation_mask`` arrays (graph
    perception stays with the caller).  Generates the n-fold angle grid in the
    reference's ``cartesian_product`` order, scans it on the GPU, keeps the
    starting structure plus every conformer with at least one rotated bond,
    TFD-prunes on the torsion quadruplets and subsamples to ``n_out``."""
    from firecode_amd.utils import cartesian_rows_at

    n_fold_angles = {2: (0, 180), 3: (0, 120, 240), 4: (0, 90, 180, 270), 6: (0, 60, 120, 180, 240, 300)}
    quads = np.array([t[:4] for t in torsions], dtype=np.int64)
    values = [n_fold_angles[int(t[4])] for t in torsions]
    n_sets = int(np.prod([len(v) for v in values], dtype=object))
    base = L.f64(coords)
    # Fingerprint every scanned conformer inside the scan kernel, TFD-prune on the (S, Q)
    # fingerprints, then generate coordinates only for the survivors -- the scan of 1.7 M
    # angle-sets would otherwise move 2 GB of conformers to the host to keep a few thousand.
    # ... and the fingerprints themselves stay on the device between the scan and the prune
    # ... and neither does the grid of angle-sets exist anywhere but on the device (107 MB at 8 x 6-fold): the
    # survivors' angles are recovered from their row numbers
    rot, keep = torsion_scan_tfd_grid(base, quads, rotation_masks, values, quads, thresh=thresh, tfd_thresh=10)
    n_new = 1 + int(np.count_nonzero(rot))  # the starting structure first, as the reference lists it (:858-861)
    if logfunction is not None:
        logfunction(f"> Group 1/1 - {len(torsions)} bonds, {[int(t[4]) for t in torsions]} n-folds, "
                    f"1 starting point = {n_sets} conformers")
    rows = np.flatnonzero(keep[1:])
    first = 1 if keep[0] else 0
    pruned = np.empty((first + len(rows),) + base.shape)  # the starting structure in front, the survivors written behind it
    if first:
        pruned[0] = base
    torsion_scan(base, quads, rotation_masks, cartesian_rows_at(values, rows), thresh=thresh, out=pruned[first:])
    if n_new > n_out:
        return np.array(most_diverse_conformers(n_out, list(pruned), seed=seed))
    return pruned


N_FOLD_ANGLES = {2: (0, 180), 3: (0, 120, 240), 4: (0, 90, 180, 270), 6: (0, 60, 120, 180, 240, 300)}


def random_csearch_core(coords, torsions, rotation_masks, n_out=100, max_tries=10000, rotations=None, thresh=1.5,
                        seed=None, order=None, return_indices=False):
    """Numeric core of ``random_csearch`` (torsion_module.py:436-571): the n-fold angle grid
    in ``cartesian_product`` order, optionally only the sets with exactly ``rotations``
    non-zero angles, SHUFFLED, then the same per-set dihedral scan as the clustered search;
    the first ``n_out`` sets that rotated at least one bond are returned, in shuffled order.

    The reference shuffles with the global unseeded NumPy RNG; here ``order`` (a permutation
    of the filtered grid) or ``seed`` (``RandomState(seed).shuffle``) fixes it -- with the same
    permutation the output is the reference's.  The stop rule is the reference's, quirk
    included: the loop ends only ON a kept set, when ``n_out`` are collected or that set's
    index equals ``max_tries``.  The grid is scanned on the GPU in chunks of shuffled sets
    until the stop rule fires."""
    from firecode_amd.utils import cartesian_product

    quads = np.array([t[:4] for t in torsions], dtype=np.int64)
    angles = cartesian_product(*[N_FOLD_ANGLES[int(t[4])] for t in torsions])
    if rotations is not None:
        angles = angles[np.count_nonzero(angles, axis=1) == rotations]
    if order is not None:
        angles = angles[np.asarray(order, dtype=np.int64)]
    else:
        rng = np.random if seed is None else np.random.RandomState(seed)
        rng.shuffle(angles)
    base = L.f64(coords)
    kept, kept_idx = [], []
    chunk = max(4 * int(n_out), 4096)
    done = False
    for lo in range(0, len(angles), chunk):
        out, rot = torsion_scan(base, quads, rotation_masks, angles[lo: lo + chunk], thresh=thresh)
        for s in np.nonzero(rot != 0)[0]:
            a = lo + int(s)
            kept.append(out[s])
            kept_idx.append(a)
            if len(kept) == n_out or a == max_tries:
                done = True
                break
        if done:
            break
    structures = np.array(kept) if kept else np.empty((0,) + base.shape)
    if return_indices:
        return structures, np.array(kept_idx, dtype=np.int64)
    return structures


# ---- the reference's own signatures (firecode/torsion_module.py:354, 436-450, 726-742, 1046) --------------------------
def _get_rotation_mask(graph, torsion):
    """firecode/torsion_module.py:354-382: the atoms that rotate with i4 (reachable from i4 once the i2-i3 bond is
    cut), i3 excluded; one entry per node of ``graph``."""
    from firecode_amd.pruner import rotation_mask

    return rotation_mask(graph, torsion)


def _get_tf_mat(structures, quadruplets):
    """firecode/torsion_module.py:1046-1053."""
    return get_tf_mat(structures, quadruplets)


def _torsion_rows(torsions, graph):
    """duck-typed ``Torsion`` objects (``.torsion`` = (i1, i2, i3, i4), ``.n_fold``) -> the numeric cores' inputs:
    (i1, i2, i3, i4, n_fold) rows and one rotation mask per torsion (this package's ``rotation_mask``)."""
    rows, masks = [], []
    for t in torsions:
        quad = tuple(int(i) for i in t.torsion)
        if int(t.n_fold) not in N_FOLD_ANGLES:
            raise L.FirecodeHipInputError(L.FC_E_INVALID, f"torsion {quad}: n_fold {t.n_fold!r} has no angle set "
                                                           "(firecode/torsion_module.py:139-145 knows 2, 3, 4, 6)")
        rows.append(quad + (int(t.n_fold),))
        masks.append(_get_rotation_mask(graph, quad))
    return rows, np.array(masks, dtype=bool)


def _log_torsions(logfunction, atoms, torsions, with_symbols):
    if logfunction is None:
        return
    logfunction("\n> Torsion list: (indices: n-fold)")
    for t, torsion in enumerate(torsions):
        ids = [int(i) for i in torsion.torsion]
        if with_symbols:
            logfunction(" {:2s} - {:21s} : {}{}{}{} : {}-fold".format(str(t), str(ids), *[atoms[i] for i in ids], torsion.n_fold))
        else:
            logfunction(" {:3} - {:21s} : {}-fold".format(t, str(ids), torsion.n_fold))
    central = sorted({int(i) for t in torsions for i in t.torsion[1:3]})
    logfunction(f"\n> Rotable bonds ids: {' '.join(str(i) for i in central)}")


def clustered_csearch(atoms, coords, torsions, graph, charge=0, mult=1, constrained_indices=None, n=100, n_out=100,
                      title="test", logfunction=print, interactive_print=True, write_torsions=False, debug=False,
                      seed=None):
    """``clustered_csearch`` with the reference's signature (firecode/torsion_module.py:726-742; called at :697-710 with
    ``Torsion`` objects from the perception step): one torsion group, its n-fold angle grid in ``cartesian_product``
    order scanned on the GPU with the clash test and the 5-degree back-off, the starting structure plus every conformer
    that rotated a bond TFD-pruned on the torsion quadruplets, ``most_diverse_conformers`` down to ``n_out``.
    ``charge`` / ``mult`` / ``n`` / ``debug`` are accepted as in the reference (it does not use the first two either;
    ``n`` only acts between torsion groups and the reference forms one group, :749).  ``write_torsions`` (VMD files)
    is outside the hot path: not supported here.  ``seed`` (extra, keyword only in practice) fixes the final draw."""
    import time

    if write_torsions:
        raise NotImplementedError("write_torsions (VMD / .xyz side files) is not part of the hot path")
    t0 = time.perf_counter()
    coords = L.f64(coords)
    _log_torsions(logfunction, atoms, torsions, with_symbols=False)
    if logfunction is not None:
        logfunction(f"\n--> Clustered CSearch on {title}\n    - {len(torsions)} torsions in 1 group - {[len(torsions)]}")
    rows, masks = _torsion_rows(torsions, graph)
    out = clustered_csearch_core(coords, rows, masks, n_out=n_out, seed=seed, logfunction=logfunction)
    if logfunction is not None:
        share = len(out) / np.prod([int(t.n_fold) for t in torsions], dtype=float)
        logfunction(f"  Selected the most diverse {len(out)} conformers, corresponding\n"
                    f"  to about {round(100 * share, 2)} % of the total conformational space - CSearch time "
                    f"{time.perf_counter() - t0:.3f} s")
    return out


def random_csearch(atoms, coords, torsions, graph, constrained_indices=None, n_out=100, max_tries=10000, rotations=None,
                   title="test", logfunction=print, interactive_print=True, write_torsions=False, seed=None, order=None):
    """``random_csearch`` with the reference's signature (firecode/torsion_module.py:436-450; called at :712-723): the
    n-fold angle grid, optionally only the sets with exactly ``rotations`` non-zero angles, shuffled, scanned until
    ``n_out`` sets rotated a bond (the reference's stop rule, ``max_tries`` quirk included).  The reference shuffles
    with the global NumPy generator; ``seed`` / ``order`` (extras) fix the permutation."""
    import time

    if write_torsions:
        raise NotImplementedError("write_torsions (VMD / .xyz side files) is not part of the hot path")
    t0 = time.perf_counter()
    coords = L.f64(coords)
    _log_torsions(logfunction, atoms, torsions, with_symbols=True)
    if logfunction is not None:
        logfunction(f"\n--> Random dihedral CSearch on {title}\n    mode 2 (random) - {len(torsions)} torsions")
    rows, masks = _torsion_rows(torsions, graph)
    out = random_csearch_core(coords, rows, masks, n_out=n_out, max_tries=max_tries, rotations=rotations, seed=seed, order=order)
    if logfunction is not None:
        share = len(out) / np.prod([int(t.n_fold) for t in torsions], dtype=float)
        logfunction(f"  Generated {len(out)} conformers, (est. {round(100 * share, 2)} % of the total conformational space) - "
                    f"CSearch time {time.perf_counter() - t0:.3f} s")
    return out
