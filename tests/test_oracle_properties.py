"""Property tests that pin the PARITY-UNPINNED oracle rows (their arithmetic
lives in the absent third-party prism_pruner): each
property must hold for any correct Kabsch / pruner, whatever its internals."""

import numpy as np
import pytest

from firecode_amd import synthetic as syn
from oracle import cpu_ref as o

rng = np.random.default_rng(99)


def test_kabsch_recovers_known_rotation():
    for _ in range(20):
        p = rng.normal(size=(17, 3))
        R = syn.random_rotation(rng)
        q = p @ R  # so that R @ q_a == p_a
        M = o.get_alignment_matrix(p, q)
        assert np.allclose(M, R, atol=1e-12)
        assert np.isclose(np.linalg.det(M), 1.0)
        r, m = o.rmsd_and_max(p, q)
        assert r < 1e-12 and m < 1e-12


def test_kabsch_never_returns_a_reflection():
    p = rng.normal(size=(12, 3))
    q = p * np.array([1, 1, -1.0])  # mirror image
    M = o.get_alignment_matrix(p, q)
    assert np.isclose(np.linalg.det(M), 1.0)
    assert o.rmsd_and_max(p, q)[0] > 0.1


def test_rmsd_is_minimal_and_symmetric():
    p = rng.normal(size=(20, 3))
    q = rng.normal(size=(20, 3))
    r, _ = o.rmsd_and_max(p, q)
    r2, _ = o.rmsd_and_max(q, p)
    assert abs(r - r2) < 1e-12
    for _ in range(50):  # no other rotation does better
        R = syn.random_rotation(rng)
        d = p - q @ R.T
        assert np.sqrt((d * d).sum() / len(d)) >= r - 1e-12


def test_center_flag_gives_translation_invariance():
    p = rng.normal(size=(15, 3))
    q = p @ syn.random_rotation(rng).T + 0.1 * rng.normal(size=p.shape)
    r0, m0 = o.rmsd_and_max(p, q, center=True)
    r1, m1 = o.rmsd_and_max(p + 3.0, q - 7.0, center=True)
    assert abs(r0 - r1) < 1e-12 and abs(m0 - m1) < 1e-12
    assert abs(o.rmsd_and_max(p + 3.0, q - 7.0, center=False)[0] - r0) > 1e-3


def test_batch_equals_literal():
    X, atoms, _ = syn.synthetic_ensemble(40, 20, seed=5)
    iu, ju = np.triu_indices(40, 1)
    rb, mb = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    for k in range(0, len(iu), 37):
        r, m = o.rmsd_and_max(X[iu[k]], X[ju[k]], center=True)
        assert abs(r - rb[k]) < 1e-12 and abs(m - mb[k]) < 1e-11


def test_two_atom_analytic_case():
    p = np.array([[0.0, 0, 0], [2.0, 0, 0]])
    q = np.array([[0.0, 0, 0], [0, 3.0, 0]])  # rotate onto x: residual is the length mismatch
    r, m = o.rmsd_and_max(p, q)
    assert abs(r - np.sqrt(0.5)) < 1e-12 and abs(m - 1.0) < 1e-12


def test_prune_literal_equals_matrix_form_and_keeps_one_per_cluster():
    X, atoms, asg = syn.synthetic_ensemble(260, 16, seed=6)
    _, lit = o.prune_by_rmsd(X, atoms, 0.5)
    S, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    assert np.array_equal(lit, o.greedy_prune_from_matrix(S))
    assert lit.sum() == len(np.unique(asg))
    assert len(np.unique(asg[lit])) == lit.sum()  # exactly one survivor per cluster


def test_prune_is_invariant_under_rigid_motion_and_ignores_hydrogens():
    X, _, _ = syn.synthetic_ensemble(120, 12, seed=7)
    atoms = np.array(["C", "H", "O", "H"] * 3)
    _, m0 = o.prune_by_rmsd(X, atoms, 0.5)
    Y = np.array([x @ syn.random_rotation(rng).T + rng.normal(size=3) for x in X])
    _, m1 = o.prune_by_rmsd(Y, atoms, 0.5)
    assert np.array_equal(m0, m1)
    Z = X.copy()
    Z[:, atoms == "H"] += rng.normal(scale=3.0, size=Z[:, atoms == "H"].shape)  # scramble hydrogens
    _, m2 = o.prune_by_rmsd(Z, atoms, 0.5)
    assert np.array_equal(m0, m2)


def test_energy_window_and_order():
    X, atoms, asg = syn.synthetic_ensemble(100, 10, seed=8)
    en = rng.uniform(0, 0.5, size=100)
    _, m = o.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0)
    assert m.sum() == len(np.unique(asg))
    # the survivor of each cluster is its HIGHEST-energy member (the last in processing order)
    for c in np.unique(asg):
        members = np.flatnonzero(asg == c)
        assert m[members[np.argmax(en[members])]]
    # a window narrower than every gap disables pruning
    _, m = o.prune_by_rmsd(X, atoms, 0.5, energies=np.arange(100.0), max_dE=0.5)
    assert m.all()


def test_moments_of_inertia_invariants():
    x = rng.normal(size=(14, 3))
    masses = rng.uniform(1, 16, size=14)
    m0 = o.get_inertia_moments(x, masses)
    m1 = o.get_inertia_moments(x @ syn.random_rotation(rng).T + 5.0, masses)
    assert np.allclose(m0, m1, rtol=1e-12)
    assert m0[0] <= m0[1] <= m0[2] and m0[0] + m0[1] >= m0[2] - 1e-9  # triangle inequality of inertia


def test_rotate_dihedral_and_dihedral_agree():
    x = syn.synthetic_skeleton(8, np.random.default_rng(3))
    tors = (1, 2, 3, 4)
    mask = np.zeros(8, dtype=bool)
    mask[4:] = True
    before = o.dihedral(x[list(tors)])
    y = o.rotate_dihedral(x, tors, 60, mask)
    after = o.dihedral(y[list(tors)])
    delta = (after - before + 180) % 360 - 180
    assert abs(abs(delta) - 60) < 1e-9
    assert np.allclose(y[:4], x[:4])  # the fixed side does not move
    d0 = np.linalg.norm(x[5] - x[4])
    assert abs(np.linalg.norm(y[5] - y[4]) - d0) < 1e-12  # rigid
    R = o.rot_mat_from_pointer(np.array([0, 0, 2.0]), 90)
    assert np.allclose(R @ np.array([1.0, 0, 0]), [0, 1, 0])


def test_align_structures_superposes_on_first():
    X, _, _ = syn.synthetic_ensemble(10, 9, seed=9, cluster_size=10)
    out = o.align_structures(X)
    for t in range(1, 10):
        assert np.sqrt(((out[t] - out[0]) ** 2).sum() / 9) < 0.15


def test_random_csearch_stop_rule_and_align_by_moi_identity():
    """random_csearch (torsion_module.py:556-558): the stop is evaluated only on a kept set;
    align_by_moi (hypermolecule_class.py:45-86): two positive diagonal arrays -> identity"""
    rng = np.random.default_rng(7)
    base = np.cumsum(rng.normal(scale=1.0, size=(12, 3)) + [1.5, 0, 0], axis=0)
    tors = np.array([[2, 3, 4, 5], [6, 7, 8, 9]])
    masks = np.zeros((2, 12), dtype=bool)
    masks[0, 5:] = True
    masks[1, 9:] = True
    grid = o.cartesian_product((0, 120, 240), (0, 120, 240))
    order = np.array([0, 3, 1, 0, 4, 8, 2, 5])  # index 0 = all-zero set: never kept
    S, idx = o.random_csearch(base, tors, masks, grid[order], n_out=100, max_tries=3)
    full, rot = o.torsion_scan(base, tors, masks, grid[order])
    kept_all = np.nonzero(rot)[0]
    assert 3 not in kept_all and len(idx) == len(kept_all)  # index 3 is not kept -> no stop at max_tries
    S2, idx2 = o.random_csearch(base, tors, masks, grid[order], n_out=100, max_tries=int(kept_all[1]))
    assert list(idx2) == list(kept_all[:2])
    S3, idx3 = o.random_csearch(base, tors, masks, grid[order], n_out=3)
    assert list(idx3) == list(kept_all[:3]) and np.array_equal(S3, full[kept_all[:3]])

    X = rng.normal(scale=2.0, size=(5, 12, 3)) + 3.0
    out = o.align_by_moi(rng.uniform(1, 16, size=12), X)
    assert np.abs(out - (X - X.mean(axis=1, keepdims=True))).max() < 1e-13


def test_oracle_kabsch_against_an_independent_implementation():
    """scipy.spatial.transform.Rotation.align_vectors solves the same problem (Wahba / Kabsch) with its
    own SVD path: rssd = sqrt(sum |a - R b|^2), so rmsd = rssd / sqrt(A); the rotation maps b onto a like
    get_alignment_matrix does (hypermolecule_class.py:77-84) -- two implementations, one answer"""
    from scipy.spatial.transform import Rotation

    for A in (4, 7, 23, 50):
        for _ in range(10):
            p = rng.normal(size=(A, 3))
            q = p @ syn.random_rotation(rng) + 0.3 * rng.normal(size=(A, 3))
            p, q = p - p.mean(axis=0), q - q.mean(axis=0)
            rot, rssd = Rotation.align_vectors(p, q)
            r, m = o.rmsd_and_max(p, q)
            assert abs(r - rssd / np.sqrt(A)) < 1e-12
            assert np.allclose(o.get_alignment_matrix(p, q), rot.as_matrix(), atol=1e-10)
            assert abs(m - np.linalg.norm(p - q @ rot.as_matrix().T, axis=1).max()) < 1e-10


def test_convention_switches_are_what_their_names_say():
    X, atoms, _ = syn.synthetic_ensemble(60, 12, seed=8)
    Xz = X - X.mean(axis=1, keepdims=True)
    r01 = o.rmsd_and_max(Xz[0], Xz[1])[0]
    twin = np.stack([X[0], X[0] + 1e-4])
    # "<" vs "<=" exactly on a pair's rmsd
    assert o.prune_by_rmsd(X[:2], atoms, r01, max_dev=10.0)[1].all()
    assert not o.prune_by_rmsd(X[:2], atoms, r01, max_dev=10.0, strict_lt=False)[1].all()
    # which member of a similar pair falls
    assert o.prune_by_rmsd(twin, atoms, 0.5)[1].tolist() == [False, True]
    assert o.prune_by_rmsd(twin, atoms, 0.5, drop="later")[1].tolist() == [True, False]
    # the max-deviation factor: one atom 1.3 A off, rmsd = 1.3 / sqrt(12) = 0.375
    spike = np.stack([X[0], X[0].copy()])
    spike[1, 0, 2] += 1.3
    assert o.prune_by_rmsd(spike, atoms, 0.7)[1].tolist() == [False, True]                      # 1.3 < 2 * 0.7
    assert o.prune_by_rmsd(spike, atoms, 0.7, maxdev_factor=1.0)[1].all()                       # ~1.1 after the fit > 0.7
    # the energy window
    assert not o.prune_by_rmsd(twin, atoms, 0.5, energies=np.array([0.0, 0.5]), max_dE=1.0)[1].all()
    assert o.prune_by_rmsd(twin, atoms, 0.5, energies=np.array([0.0, 1.0]), max_dE=1.0)[1].all()
    assert not o.prune_by_rmsd(twin, atoms, 0.5, energies=np.array([0.0, 1.0]), max_dE=1.0, window_strict=False)[1].all()
    # no threshold given: the default
    assert np.array_equal(o.prune_by_rmsd(X, atoms)[1], o.prune_by_rmsd(X, atoms, o.CONVENTIONS["default_max_rmsd"])[1])
