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
        return ok.view(np.bool_), counts, poses
    return ok.view(np.bool_), counts


def _mol_args(coords, reactive, pivots):
    X = L.f64(coords)
    r = L.i64(reactive).reshape(-1)
    pv = L.f64(pivots)
    if X.ndim != 3 or X.shape[2] != 3 or pv.shape != (X.shape[0], 2, 3) or r.shape[0] not in (1, 2):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "coords (n, A, 3), reactive (1|2,), pivots (n, 2, 3) expected")
    return X, r, np.ascontiguousarray(pv[:, 0]), np.ascontiguousarray(pv[:, 1])


def embed_mol_transforms(coords, reactive, pivots, mol, angles):
    """Per-molecule pose transforms of the bimolecular cyclical embed
    (embeds.py:649-709): R (n, 2, na, 3, 3), t (n, 2, na, 3) over (conformer,
    orientation, step angle).  pivots[c] = (start, end) of the conformer's pivot."""
    X, r, ps, pe = _mol_args(coords, reactive, pivots)
    ang = L.f64(angles).reshape(-1)
    n, na = X.shape[0], ang.shape[0]
    R = np.empty((n, 2, na, 3, 3))
    t = np.empty((n, 2, na, 3))
    L.call("fc_embed_mol_transforms", L.pf(X), n, X.shape[1], L.pi(r), r.shape[0], L.pf(ps), L.pf(pe),
           int(mol), L.pf(ang), na, L.pf(R), L.pf(t))
    return R, t


def embed_grid_clash(m1, reactive1, pivots1, m2, reactive2, pivots2, angles1, angles2=None,
                     thresh=1.5, max_clashes=0, return_counts=False):
    """Every pose of the bimolecular rigid embed at once (the loop of
    embeds.py:597-722 for one pivot per conformer, up to the clash test).
    Returns pass (n2, n1, 2, na2, na1) bool -- index [c2, c1, o, a2, a1], the
    reference's iteration order flattened -- and the kernel time in ms."""
    import ctypes as C

    X1, r1, ps1, pe1 = _mol_args(m1, reactive1, pivots1)
    X2, r2, ps2, pe2 = _mol_args(m2, reactive2, pivots2)
    a1 = L.f64(angles1).reshape(-1)
    a2 = a1 if angles2 is None else L.f64(angles2).reshape(-1)
    shape = (X2.shape[0], X1.shape[0], 2, a2.shape[0], a1.shape[0])
    ok = np.zeros(shape, dtype=np.uint8)
    counts = np.zeros(shape, dtype=np.int32) if return_counts else None
    ms = C.c_double(0)
    L.call("fc_embed_grid_clash", L.pf(X1), X1.shape[0], X1.shape[1], L.pi(r1), r1.shape[0], L.pf(ps1), L.pf(pe1),
           L.pf(X2), X2.shape[0], X2.shape[1], L.pi(r2), r2.shape[0], L.pf(ps2), L.pf(pe2),
           L.pf(a1), a1.shape[0], L.pf(a2), a2.shape[0], float(thresh), int(max_clashes), L.pb(ok),
           None if counts is None else counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms))
    if return_counts:
        return ok.view(np.bool_), counts, ms.value
    return ok.view(np.bool_), ms.value


def embed_grid_poses(m1, reactive1, pivots1, m2, reactive2, pivots2, angles1, angles2=None,
                     thresh=1.5, max_clashes=0, rmsd_thr=1.0):
    """The pose selection of ``_fast_bimol_rigid_cyclical_embed`` (embeds.py:597-727,
    one pivot per conformer): clash test, then -- inside every (conformer pair,
    orientation) group, in angle order -- ``rmsd_similarity`` against the poses of
    the group kept so far.  Returns (accept, clash_pass), both (n2, n1, 2, na2, na1)
    bool; ``accept.reshape(-1)`` is in the reference's iteration order."""
    X1, r1, ps1, pe1 = _mol_args(m1, reactive1, pivots1)
    X2, r2, ps2, pe2 = _mol_args(m2, reactive2, pivots2)
    a1 = L.f64(angles1).reshape(-1)
    a2 = a1 if angles2 is None else L.f64(angles2).reshape(-1)
    shape = (X2.shape[0], X1.shape[0], 2, a2.shape[0], a1.shape[0])
    ok = np.zeros(shape, dtype=np.uint8)
    acc = np.zeros(shape, dtype=np.uint8)
    L.call("fc_embed_grid_dedupe", L.pf(X1), X1.shape[0], X1.shape[1], L.pi(r1), r1.shape[0], L.pf(ps1), L.pf(pe1),
           L.pf(X2), X2.shape[0], X2.shape[1], L.pi(r2), r2.shape[0], L.pf(ps2), L.pf(pe2),
           L.pf(a1), a1.shape[0], L.pf(a2), a2.shape[0], float(thresh), int(max_clashes), float(rmsd_thr),
           L.pb(ok), L.pb(acc))
    return acc.view(np.bool_), ok.view(np.bool_)


def string_embed_poses(m1, centers1, orbvecs1, m2, centers2, orbvecs2, angles, quadruplets,
                       thresh=1.5, max_clashes=0, tfd_thresh=10):
    """Pose loop of ``string_embed`` (firecode/embeds.py:51-158): molecule 1 fixed,
    molecule 2 oriented orbital-against-orbital and spun by every angle; clash
    test; sequential torsion-fingerprint novelty filter.  centers*/orbvecs*:
    (n_conf, K, 3) orbital centres / vectors of the reactive atom.
    Returns (poses (n_accepted, A1+A2, 3), accept (P,) bool, clash_pass (P,) bool),
    P in the reference's iteration order."""
    X1, X2 = L.f64(m1), L.f64(m2)
    c1, v1, c2, v2 = (L.f64(a) for a in (centers1, orbvecs1, centers2, orbvecs2))
    ang = L.f64(angles).reshape(-1)
    quads = L.i64(quadruplets).reshape(-1, 4)
    n1, n2, K1, K2, nA = X1.shape[0], X2.shape[0], c1.shape[1], c2.shape[1], ang.shape[0]
    if c1.shape != (n1, K1, 3) or v1.shape != c1.shape or c2.shape != (n2, K2, 3) or v2.shape != c2.shape:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "centers/orbvecs must be (n_conf, K, 3)")
    P = n1 * n2 * K1 * K2 * nA
    ok = np.zeros(P, dtype=np.uint8)
    acc = np.zeros(P, dtype=np.uint8)
    R2 = np.empty((P, 3, 3))
    t2 = np.empty((P, 3))
    L.call("fc_string_embed", L.pf(X1), n1, X1.shape[1], L.pf(c1), L.pf(v1), K1, L.pf(X2), n2, X2.shape[1],
           L.pf(c2), L.pf(v2), K2, L.pf(ang), nA, L.pi(quads), quads.shape[0], float(thresh), int(max_clashes),
           float(tfd_thresh), L.pb(ok), L.pb(acc), L.pf(R2), L.pf(t2))
    acc = acc.view(np.bool_)
    sel = np.flatnonzero(acc)
    ci = sel // (K1 * K2 * nA)
    moved = rototranslate(X2[ci // n1], R2[sel], t2[sel]) if len(sel) else np.zeros((0, X2.shape[1], 3))
    poses = np.concatenate([X1[ci % n1], moved], axis=1)
    return poses, acc, ok.view(np.bool_)


def _tri_mol(m):
    get = m.get if isinstance(m, dict) else (lambda k: getattr(m, k))
    return (L.f64(get("coords")), L.i64(get("reactive_indices")), get("pivots"), dict(get("reactive_cumnums")))


def cyclical_embed_trimolecular(mols, systematic_angles, pairings_table=None, internal_constraints=(),
                                clash_thresh=1.5, max_clashes=0, rmsd_thr=1.0, return_details=False):
    """``cyclical_embed`` for three molecules (firecode/embeds.py:409-585).

    ``mols``: three objects (or dicts) with ``coords`` (n_conf, A, 3), ``reactive_indices``
    (two atom indices), ``pivots[conf]`` = list of ``(start_xyz, end_xyz, start_cumnum,
    end_cumnum)`` (FIRECODE's Pivot: orbital centres and the cumulative numbers of their
    atoms) and ``reactive_cumnums`` = {atom index: cumnum} (``reactive_atoms_classes_dict[0]``).
    ``systematic_angles``: (S, 3) step angles in degrees (``embedder.systematic_angles``).

    The host enumerates (conformer triple, pivot triple) jobs in the reference's order and
    computes their O(1) set-up (norms, ``polygonize``, ``_get_directions``, the pairings
    filter, the reactive-pair table); ``_adjust_directions``, the S poses of each of the 8
    orientations, their clash test and the sequential ``rmsd_similarity`` filter run on the
    GPU (``fc_embed_trimolecular``).  Returns ``(poses (P, A1+A2+A3, 3), constrained_indices
    (P, 3, 2))`` in the reference's order; with ``return_details`` also a dict with the
    per-group ``jobs``, ``directions``, ``passed`` and ``accepted`` arrays."""
    import ctypes as C

    from firecode_amd import host_helpers as hh
    from firecode_amd.utils import cartesian_product

    if len(mols) != 3:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "three molecules expected")
    coords, reactive, pivots, cumnums = zip(*[_tri_mol(m) for m in mols])
    ang = L.f64(systematic_angles).reshape(-1, 3)
    S = ang.shape[0]
    cum2atom = {}
    for m in range(3):
        for index, cumnum in cumnums[m].items():
            cum2atom[int(cumnum)] = (m, int(index))
    wanted = list(pairings_table.values()) if pairings_table else []
    internal = [tuple(c) for c in np.asarray(internal_constraints).tolist()] if len(internal_constraints) else []

    jobs, conf, ps, pe, vecs, dirs0, run, rtab, norms_all, ids_all = [], [], [], [], [], [], [], [], [], []
    conf_indices = cartesian_product(*[np.arange(len(c)) for c in coords])
    for conf_ids in conf_indices:
        piv_idx = cartesian_product(*[np.arange(len(pivots[i][conf_ids[i]])) for i in range(3)])
        for pi in piv_idx:
            piv = [pivots[i][conf_ids[i]][pi[i]] for i in range(3)]
            start = np.array([np.asarray(p[0], float) for p in piv])
            end = np.array([np.asarray(p[1], float) for p in piv])
            norms = np.linalg.norm(start - end, axis=1)
            if not all(norms[i] < norms[i - 1] + norms[i - 2] for i in (0, 1, 2)):
                continue  # no triangle with these pivots (embeds.py:455-458)
            polygon = hh.polygonize(norms)
            d0 = hh.triangle_directions(norms)  # may nudge norms[0], as the reference does
            cum_ids = [(int(p[2]), int(p[3])) for p in piv]
            run_j, rt_j, ids_j = np.zeros(8, np.uint8), np.zeros((8, 3, 3), np.int64), []
            for v in range(8):
                ids = hh.cyclical_reactive_indices_tri(cum_ids, v)
                ids_j.append(ids)
                ok = (not wanted) or all((tuple(p) in ids) or (tuple(p) in internal) for p in wanted)
                run_j[v] = ok
                if ok:  # r[m, k]: reactive atom of molecule m that faces molecule k (:338-352)
                    for c0, c1 in ids:
                        (ma, ia), (mb, ib) = cum2atom[c0], cum2atom[c1]
                        rt_j[v, ma, mb] = ia
                        rt_j[v, mb, ma] = ib
            jobs.append((tuple(int(c) for c in conf_ids), tuple(int(p) for p in pi)))
            conf.append(conf_ids)
            ps.append(start)
            pe.append(end)
            vecs.append(polygon)
            dirs0.append(d0)
            run.append(run_j)
            rtab.append(rt_j)
            norms_all.append(norms)
            ids_all.append(ids_j)
    A = [c.shape[1] for c in coords]
    n_atoms = sum(A)
    J = len(jobs)
    if J == 0:
        empty = (np.empty((0, n_atoms, 3)), np.empty((0, 3, 2), dtype=np.int64))
        return empty + ({"jobs": []},) if return_details else empty
    # distinct step angles per molecule
    uniq = [np.unique(ang[:, i], return_inverse=True) for i in range(3)]
    U = max(len(u[0]) for u in uniq)
    ua = np.zeros((3, U))
    for i in range(3):
        ua[i, : len(uniq[i][0])] = uniq[i][0]
    aidx = np.ascontiguousarray(np.stack([u[1] for u in uniq], axis=1).astype(np.int32))
    conf_a, ps_a, pe_a = L.i64(np.array(conf)), L.f64(np.array(ps)), L.f64(np.array(pe))
    vecs_a, d0_a, run_a = L.f64(np.array(vecs)), L.f64(np.array(dirs0)), L.u8(np.array(run))
    rt_a, nm_a = L.i64(np.array(rtab)), L.f64(np.array(norms_all))
    dirs_out = np.empty((J, 8, 3, 3))
    Rt = np.empty((J, 8, 3, U, 12))
    passed = np.zeros((J, 8, S), dtype=np.uint8)
    accepted = np.zeros((J, 8, S), dtype=np.uint8)
    cptr = (C.POINTER(C.c_double) * 3)(*[L.pf(c) for c in coords])
    rptr = (C.POINTER(C.c_int64) * 3)(*[L.pi(r) for r in reactive])
    nconf = L.i64([len(c) for c in coords])
    natm = L.i64(A)
    nreact = L.i64([len(r) for r in reactive])
    L.call("fc_embed_trimolecular", cptr, L.pi(nconf), L.pi(natm), rptr, L.pi(nreact), J, L.pi(conf_a),
           L.pf(ps_a), L.pf(pe_a), L.pf(vecs_a), L.pf(d0_a), L.pb(run_a), L.pi(rt_a), L.pf(nm_a), L.pf(ua), U,
           aidx.ctypes.data_as(C.POINTER(C.c_int32)), S, float(clash_thresh), int(max_clashes), float(rmsd_thr),
           L.pf(dirs_out), L.pf(Rt), L.pb(passed), L.pb(accepted))
    # accepted poses, in loop order: molecule i = R x + t with (R, t) of its step angle
    jj, vv, ss = np.nonzero(accepted)
    parts = []
    for i in range(3):
        rt = Rt[jj, vv, i, aidx[ss, i]]
        parts.append(rototranslate(coords[i][conf_a[jj, i]], rt[:, :9].reshape(-1, 3, 3), rt[:, 9:])
                     if len(jj) else np.empty((0, A[i], 3)))
    poses = np.concatenate(parts, axis=1)
    constrained = np.array([ids_all[j][v] for j, v in zip(jj, vv)], dtype=np.int64).reshape(-1, 3, 2)
    if return_details:
        return poses, constrained, {"jobs": jobs, "directions": dirs_out, "passed": passed.astype(bool),
                                    "accepted": accepted.astype(bool), "run": np.array(run).astype(bool)}
    return poses, constrained


def cyclical_embed_bimolecular(mols, systematic_angles, pairings_table=None, internal_constraints=(),
                               clash_thresh=1.5, max_clashes=0, rmsd_thr=1.0, max_norm_delta=10.0):
    """``_fast_bimol_rigid_cyclical_embed`` (firecode/embeds.py:588-750) for two molecules
    given like in ``cyclical_embed_trimolecular`` (any number of pivots per conformer).

    Every (conformer, pivot) of a molecule becomes one entry of its table, the whole pose grid
    -- clash test and the in-group ``rmsd_similarity`` filter -- is ONE library call
    (``fc_embed_grid_dedupe``); the host then walks the reference's loop order (conformer pairs,
    pivot pairs, orientation), drops what the reference skips (pivot norms more than
    ``max_norm_delta`` apart, :624-626; orientations that do not realise the requested
    pairings, :640-643) and builds the accepted poses.  ``systematic_angles`` must be the
    embedder's grid (embedder.py:1090-1098).  Returns (poses, constrained_indices (P, 2, 2))."""
    from firecode_amd import host_helpers as hh

    if len(mols) != 2:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "two molecules expected")
    coords, reactive, pivots, _ = zip(*[_tri_mol(m) for m in mols])
    ang = L.f64(systematic_angles).reshape(-1, 2)
    ua = [np.unique(ang[:, i]) for i in range(2)]
    grid = np.stack([np.tile(ua[0], len(ua[1])), np.repeat(ua[1], len(ua[0]))], axis=1)
    if grid.shape != ang.shape or not np.array_equal(grid, ang):
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "systematic_angles is not the cartesian_product grid of the embedder")
    # tables of (conformer, pivot) entries
    vconf, vpiv, vidx = [], [], []
    for m in range(2):
        conf_of, piv_of, index = [], [], []
        for c in range(len(coords[m])):
            row = []
            for p, pv in enumerate(pivots[m][c]):
                row.append(len(conf_of))
                conf_of.append(c)
                piv_of.append(pv)
            index.append(row)
        vconf.append(np.array(conf_of, dtype=np.int64))
        vpiv.append(piv_of)
        vidx.append(index)
    A = [c.shape[1] for c in coords]
    if min(len(v) for v in vconf) == 0:
        return np.empty((0, sum(A), 3)), np.empty((0, 2, 2), dtype=np.int64)
    vcoords = [np.ascontiguousarray(coords[m][vconf[m]]) for m in range(2)]
    vpv = [np.array([[np.asarray(p[0], float), np.asarray(p[1], float)] for p in vpiv[m]]) for m in range(2)]
    norms = [np.linalg.norm(vpv[m][:, 0] - vpv[m][:, 1], axis=1) for m in range(2)]
    acc, _ = embed_grid_poses(vcoords[0], reactive[0], vpv[0], vcoords[1], reactive[1], vpv[1], ua[0], ua[1],
                              thresh=clash_thresh, max_clashes=max_clashes, rmsd_thr=rmsd_thr)
    wanted = [tuple(p) for p in pairings_table.values()] if pairings_table else []
    internal = [tuple(c) for c in np.asarray(internal_constraints).tolist()] if len(internal_constraints) else []
    sel_vb, sel_va, sel_o, sel_a2, sel_a1, cons = [], [], [], [], [], []
    n0, n1 = len(coords[0]), len(coords[1])
    for c1 in range(n1):          # cartesian_product: the second molecule's index is the slowest
        for c0 in range(n0):
            for p1 in range(len(vidx[1][c1])):
                for p0 in range(len(vidx[0][c0])):
                    va, vb = vidx[0][c0][p0], vidx[1][c1][p1]
                    if abs(norms[0][va] - norms[1][vb]) > max_norm_delta:
                        continue
                    cum = [(int(vpiv[0][va][2]), int(vpiv[0][va][3])), (int(vpiv[1][vb][2]), int(vpiv[1][vb][3]))]
                    for orient in (0, 1):
                        ids = hh.cyclical_reactive_indices(cum[0], cum[1], orient)
                        if wanted and not all((p in ids) or (p in internal) for p in wanted):
                            continue
                        a2i, a1i = np.nonzero(acc[vb, va, orient])
                        sel_vb.extend([vb] * len(a1i))
                        sel_va.extend([va] * len(a1i))
                        sel_o.extend([orient] * len(a1i))
                        sel_a2.extend(a2i.tolist())
                        sel_a1.extend(a1i.tolist())
                        cons.extend([ids] * len(a1i))
    if not sel_va:
        return np.empty((0, sum(A), 3)), np.empty((0, 2, 2), dtype=np.int64)
    R0, t0 = embed_mol_transforms(vcoords[0], reactive[0], vpv[0], 0, ua[0])
    R1, t1 = embed_mol_transforms(vcoords[1], reactive[1], vpv[1], 1, ua[1])
    va, vb, oo = np.array(sel_va), np.array(sel_vb), np.array(sel_o)
    a1i, a2i = np.array(sel_a1), np.array(sel_a2)
    part0 = rototranslate(vcoords[0][va], R0[va, oo, a1i], t0[va, oo, a1i])
    part1 = rototranslate(vcoords[1][vb], R1[vb, oo, a2i], t1[vb, oo, a2i])
    return np.concatenate([part0, part1], axis=1), np.array(cons, dtype=np.int64).reshape(-1, 2, 2)


# ---- the reference's own signatures: string_embed(embedder), cyclical_embed(embedder, max_norm_delta) -------------------
class ZeroCandidatesError(Exception):
    """firecode/errors.py:24-27: raised when an embed finds no pose (the reference's callers catch it by this name;
    a run that has the reference installed gets ITS class, so that ``except ZeroCandidatesError`` keeps working)."""


def _zero_candidates_error():
    try:
        from firecode.errors import ZeroCandidatesError as ref_error  # the caller's own class when FIRECODE is importable

        return ref_error
    except Exception:  # noqa: BLE001 -- not installed (tests, stand-alone use)
        return ZeroCandidatesError


def _embedder_mols(embedder):
    """The numeric cores' view of ``embedder.objects`` (Hypermolecule objects, firecode/hypermolecule_class.py):
    coordinates, the two reactive indices, per conformer the pivots as (start, end, cumnum of the start atom, cumnum of
    the end atom) (``Pivot``, :300-336) and the cumulative numbers of the reactive atoms."""
    mols = []
    for mol in embedder.objects:
        pivots = [[(p.start, p.end, p.start_atom.cumnum, p.end_atom.cumnum) for p in mol.pivots[c]]
                  for c in range(len(mol.coords))]
        cum = {int(i): int(ra.cumnum) for i, ra in mol.reactive_atoms_classes_dict[0].items()}
        mols.append({"coords": mol.coords, "reactive_indices": mol.reactive_indices, "pivots": pivots,
                     "reactive_cumnums": cum})
    return mols


def cyclical_embed(embedder, max_norm_delta=5.0):
    """``cyclical_embed`` with the reference's signature (firecode/embeds.py:180; two molecules: the fast rigid form
    :588-750, three: :409-585): reads ``embedder.objects`` (coords, reactive_indices, pivots, reactive_atoms_classes_dict),
    ``systematic_angles``, ``pairings_table``, ``internal_constraints``, ``options.clash_thresh``; sets
    ``embedder.constrained_indices``; returns the poses; raises ``ZeroCandidatesError`` when there are none."""
    mols = _embedder_mols(embedder)
    if hasattr(embedder, "log"):
        embedder.log(f"\n--> Performing {getattr(embedder, 'embed', 'cyclical')} embed ({getattr(embedder, 'candidates', '?')} candidates)")
    pairings = getattr(embedder, "pairings_table", None) or None
    internal = getattr(embedder, "internal_constraints", ())
    thresh = embedder.options.clash_thresh
    if len(mols) == 2:
        poses, constrained = cyclical_embed_bimolecular(mols, embedder.systematic_angles, pairings, internal,
                                                        clash_thresh=thresh, max_norm_delta=max_norm_delta)
    elif len(mols) == 3:
        poses, constrained = cyclical_embed_trimolecular(mols, embedder.systematic_angles, pairings, internal,
                                                         clash_thresh=thresh)
    else:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "cyclical embed takes two or three molecules")
    embedder.constrained_indices = np.asarray(constrained)
    if len(poses) == 0:
        msg = ("\n--> Cyclical embed did not find any suitable disposition of molecules.\n"
               "    This is probably because one molecule has two reactive centers at a great distance,\n"
               "    preventing the other two molecules from forming a closed, cyclical structure.")
        if hasattr(embedder, "log"):
            embedder.log(msg, p=False)
        raise _zero_candidates_error()(msg)
    return poses


def string_embed(embedder):
    """``string_embed`` with the reference's signature (firecode/embeds.py:51-158): two molecules, the first reactive
    atom of each, every (conformer pair, orbital pair, step angle); clash test at ``options.clash_thresh``; sequential
    torsion-fingerprint novelty filter on the quadruplets of the joined bond graph (:109; this package's
    ``torsion_perception.get_quadruplets`` over the two molecular graphs + the forming bond).  Sets
    ``embedder.constrained_indices`` (:161-178)."""
    import networkx as nx

    from firecode_amd.torsion_perception import get_quadruplets

    if len(embedder.objects) != 2:
        raise L.FirecodeHipInputError(L.FC_E_INVALID, "string embed takes two molecules")
    mol1, mol2 = embedder.objects
    if hasattr(embedder, "log"):
        embedder.log(f"\n--> Performing string embed ({getattr(embedder, 'candidates', '?')} candidates)")
    pair = [int(mol1.reactive_indices[0]), int(mol2.reactive_indices[0] + embedder.ids[0])]
    # the joined bond graph as get_sum_graph builds it (firecode/graph_manipulations.py:117-141): the first graph, then the
    # second one's EDGES shifted by the node count (its nodes appear in edge order -- the torsion list follows the graph's
    # iteration order), then the forming bond, the element symbols renumbered cumulatively
    joined = nx.Graph(mol1.graph)
    shift = joined.number_of_nodes()
    for e1, e2 in mol2.graph.edges():
        joined.add_edge(e1 + shift, e2 + shift)
    joined.add_edge(*pair)
    symbols = list(nx.get_node_attributes(mol1.graph, "atoms").values()) + list(nx.get_node_attributes(mol2.graph, "atoms").values())
    nx.set_node_attributes(joined, dict(enumerate(symbols)), "atoms")
    quadruplets = get_quadruplets(joined)

    def orbitals(mol):
        ras = [mol.get_r_atoms(c)[0] for c in range(len(mol.coords))]
        return np.array([ra.center for ra in ras], dtype=float), np.array([ra.orb_vecs for ra in ras], dtype=float)

    c1, v1 = orbitals(mol1)
    c2, v2 = orbitals(mol2)
    poses, _, _ = string_embed_poses(mol1.coords, c1, v1, mol2.coords, c2, v2, np.asarray(embedder.systematic_angles, dtype=float),
                                     quadruplets, thresh=embedder.options.clash_thresh)
    if len(poses) == 0:
        msg = ("\n--> Cyclical embed did not find any suitable disposition of molecules.\n"
               "    This is probably because the two molecules cannot find a correct interlocking pose.\n"
               "    Try expanding the conformational space with the firecode_search> operator or see the SHRINK keyword.")
        if hasattr(embedder, "log"):
            embedder.log(msg, p=False)
        raise _zero_candidates_error()(msg)
    embedder.constrained_indices = np.array([[pair] for _ in range(len(poses))])
    return poses
