"""Consumes tests/golden/prism_v1.npz -- the outputs of the real third-party ``prism_pruner``
(tests/golden/make_golden_prism.py) -- when it exists; until someone has run that script where
the package is installed these tests SKIP and the third-party rows stay "parity unpinned"
(DESIGN.md section 2).  CPU tests check the oracle; the gpu test checks the HIP path."""

import os

import numpy as np
import pytest

from oracle import cpu_ref as o

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prism_v1.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="prism_v1.npz not generated yet: parity unpinned")
TOL = 1e-10


@pytest.fixture(scope="module")
def P():
    return np.load(PATH, allow_pickle=False)


def test_oracle_rmsd_and_max_against_the_package(P):
    X = P["X"]
    iu, ju = P["rm_pairs"].T
    got = np.array([o.rmsd_and_max(X[a], X[b], center=True) for a, b in zip(iu, ju)])
    assert np.abs(got - P["rm_center_true"]).max() < TOL


def test_which_conventions_reproduce_the_package_masks(P):
    """Every combination of the named switches against the package's masks: exactly the combinations
    that reproduce ALL of them are printed; the defaults must be among them."""
    X, atoms, en = P["X"], P["atoms"], P["energies"]
    import itertools

    good = []
    for strict, factor, drop, wstrict in itertools.product((True, False), (1.0, 1.5, 2.0, 3.0), ("earlier", "later"),
                                                           (True, False)):
        kw = dict(strict_lt=strict, maxdev_factor=factor, drop=drop, window_strict=wstrict)
        ok = all(np.array_equal(o.prune_by_rmsd(X, atoms, t, **kw)[1], P[f"prune_rmsd_clustered_{t}"]) for t in (0.25, 0.5, 1.0))
        ok = ok and np.array_equal(o.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0, **kw)[1], P["prune_rmsd_energies_1.0"])
        ok = ok and np.array_equal(o.prune_by_rmsd(P["prune_rmsd_spike_in"], atoms, 0.7, **kw)[1], P["prune_rmsd_spike_0.7"])
        ok = ok and np.array_equal(o.prune_by_rmsd(X[:2], atoms, float(P["prune_rmsd_tie_thr"]), **kw)[1], P["prune_rmsd_tie_pair"])
        if ok:
            good.append(kw)
    print("conventions that reproduce prism_pruner:", good)
    assert dict(strict_lt=o.CONVENTIONS["strict_lt"], maxdev_factor=o.CONVENTIONS["maxdev_factor"],
                drop=o.CONVENTIONS["drop"], window_strict=o.CONVENTIONS["window_strict"]) in good


def test_oracle_algebra_against_the_package(P):
    assert np.abs(np.array([o.dihedral(p) for p in P["dihedral_in"]]) - P["dihedral_out"]).max() < TOL
    got = np.array([o.rot_mat_from_pointer(a, t) for a, t in zip(P["rmfp_axis"], P["rmfp_angle"])])
    assert np.abs(got - P["rmfp_out"]).max() < TOL
    got = np.array([o.rotate_dihedral(P["rd_base"], tuple(P["rd_torsion"]), a, P["rd_mask"]) for a in P["rd_angles"]])
    assert np.abs(got - P["rd_out"]).max() < TOL


@pytest.mark.gpu
def test_hip_path_against_the_package(fc, P):
    X, atoms = P["X"], P["atoms"]
    iu, ju = P["rm_pairs"].T
    r, m = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True)
    assert np.abs(np.stack([r, m], 1) - P["rm_center_true"]).max() < TOL
    for t in (0.25, 0.5, 1.0):
        assert np.array_equal(fc.pruner.prune_by_rmsd(X, atoms, t)[1], P[f"prune_rmsd_clustered_{t}"])
        assert np.array_equal(fc.pruner.prune_by_rmsd(P["Xc"], atoms, t)[1], P[f"prune_rmsd_continuous_{t}"])
    assert np.array_equal(fc.pruner.prune_by_moment_of_inertia(X, atoms)[1], P["prune_moi"])
