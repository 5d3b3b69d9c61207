"""host_helpers against the reference's own outputs (golden) -- CPU only."""

import numpy as np

from firecode_amd import host_helpers as hh
from oracle import cpu_ref as o


def test_polygonize_and_rmv_golden(golden):
    assert np.array_equal(hh.polygonize(golden["poly2_in"]), golden["poly2_out"])
    assert np.array_equal(hh.polygonize(golden["poly3_in"]), golden["poly3_out"])
    out = np.array([hh.rotation_matrix_from_vectors(a, b) for a, b in zip(golden["rmv_v1"], golden["rmv_v2"])])
    assert np.array_equal(out, golden["rmv_out"])


def test_rot_mat_and_angles():
    R = hh.rot_mat_from_pointer([0, 0, 3.0], 90)
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0])
    assert np.array_equal(R, o.rot_mat_from_pointer(np.array([0, 0, 3.0]), 90))
    assert np.allclose(hh.rotation_matrix_from_vectors(np.array([1.0, 0, 0]), np.array([-1.0, 0, 0])) @ [1, 0, 0], [-1, 0, 0])
    ang = hh.systematic_angles(5, 45)
    ref = (o.cartesian_product(range(6), range(6)) * 2 * 45 / 5 - 45)
    assert np.array_equal(np.unique(ref), ang) and len(ang) == 6
    assert hh.cyclical_reactive_indices([3, 9], [20, 25], 0) == [(3, 20), (9, 25)]
    assert hh.cyclical_reactive_indices([3, 9], [20, 25], 1) == [(3, 25), (9, 20)]


def test_triangle_helpers_against_the_restatement():
    """triangle_directions / cyclical_reactive_indices_tri (product host code) against the
    oracle's literal _get_directions / _get_cyclical_reactive_indices; geometric sanity"""
    from oracle import cyclical_ref as cy

    rng = np.random.default_rng(5)
    for _ in range(200):
        n = rng.uniform(1.5, 4.0, size=3)
        if not all(n[i] < n[i - 1] + n[i - 2] for i in (0, 1, 2)):
            continue
        d = hh.triangle_directions(n.copy())
        assert np.array_equal(d, cy.get_directions(n.copy()))
        assert np.allclose(np.linalg.norm(d, axis=1), 1.0) and np.all(d[:, 2] == 0)
    # acute triangle: every direction points from the side's midpoint to the circumcentre
    n = np.array([3.0, 3.2, 2.9])
    d = hh.triangle_directions(n.copy())
    poly = hh.polygonize(n)[0]
    mids = poly.mean(axis=1)
    # lines mid_k + t d_k meet in one point
    A = np.array([[d[0, 0], -d[1, 0]], [d[0, 1], -d[1, 1]]])
    t = np.linalg.solve(A, (mids[1] - mids[0])[:2])
    centre = mids[0][:2] + t[0] * d[0][:2]
    verts = poly[:, 0, :2]
    assert np.allclose(np.linalg.norm(verts - centre, axis=1), np.linalg.norm(verts[0] - centre))
    # right triangle: the reference nudges norms[0]
    n = np.array([3.0, 4.0, 5.0])
    d = hh.triangle_directions(n)
    assert n[0] == 3.0 + 1e-5 and np.isfinite(d).all()
    piv = [cy.Pivot(np.zeros(3), np.ones(3), a, b) for a, b in ((2, 7), (13, 19), (25, 31))]
    for v in range(8):
        assert hh.cyclical_reactive_indices_tri([(2, 7), (13, 19), (25, 31)], v) == cy.get_cyclical_reactive_indices(piv, v)


def test_rotation_mask_golden(golden):
    """pruner.rotation_mask against the reference's own _get_rotation_mask output"""
    import networkx as nx

    from firecode_amd.pruner import rotation_mask

    graph = nx.Graph([tuple(int(x) for x in e) for e in golden["rotmask_edges"]])
    out = np.array([rotation_mask(graph, t, graph.number_of_nodes()) for t in golden["rotmask_torsions"]])
    assert np.array_equal(out, golden["rotmask_out"])
    assert graph.number_of_edges() == len(golden["rotmask_edges"])  # the graph is restored


def test_cyclical_reactive_indices_golden(golden):
    """host helpers (and the oracle's copies) against the reference's _get_cyclical_reactive_indices"""
    from oracle import cyclical_ref as cy

    c2, c3 = golden["cri_cum2"].tolist(), golden["cri_cum3"].tolist()
    for n in range(2):
        assert np.array_equal(hh.cyclical_reactive_indices(c2[0], c2[1], n), golden["cri_out2"][n])
        piv = [cy.Pivot(np.zeros(3), np.ones(3), a, b) for a, b in c2]
        assert np.array_equal(cy.get_cyclical_reactive_indices_bimol(piv, n), golden["cri_out2"][n])
    for n in range(8):
        assert np.array_equal(hh.cyclical_reactive_indices_tri([tuple(c) for c in c3], n), golden["cri_out3"][n])
        piv = [cy.Pivot(np.zeros(3), np.ones(3), a, b) for a, b in c3]
        assert np.array_equal(cy.get_cyclical_reactive_indices(piv, n), golden["cri_out3"][n])


def test_most_diverse_conformers_is_the_reference_draw():
    """a22 (firecode/torsion_module.py:574-586): all structures when there are at most n; otherwise
    np.sort(np.random.choice(N, size=n)) -- with replacement, so indices may repeat -- from the global
    RNG (unseeded in the reference), or from RandomState(seed) when a seed is given"""
    import numpy as np

    from firecode_amd.torsion_module import most_diverse_conformers

    S = np.arange(40 * 6 * 3, dtype=float).reshape(40, 6, 3)
    out = most_diverse_conformers(50, S)
    assert len(out) == 40 and all(np.array_equal(a, b) for a, b in zip(out, S))
    assert len(most_diverse_conformers(40, list(S))) == 40
    for seed in (0, 7, 123):
        idx = np.sort(np.random.RandomState(seed).choice(40, size=12))
        out = most_diverse_conformers(12, S, seed=seed)
        assert len(out) == 12 and all(np.array_equal(a, S[i]) for a, i in zip(out, idx))
    np.random.seed(99)  # the reference's own path: the global RNG
    idx = np.sort(np.random.choice(40, size=5))
    np.random.seed(99)
    out = most_diverse_conformers(5, S)
    assert all(np.array_equal(a, S[i]) for a, i in zip(out, idx))


def test_cartesian_product_native_equals_numpy(golden):
    """firecode_amd.utils.cartesian_product: integer and floating grids are written by the library's host code
    (fc_cartesian_product_*), everything else by the reference's NumPy expression -- same rows, order, shape and
    dtype as np.stack(np.meshgrid(*arrays), -1).reshape(-1, T) either way; golden vectors of the reference too"""
    from firecode_amd.utils import cartesian_product

    def ref(*arrays):
        a = [np.asarray(x) for x in arrays]
        return np.stack(np.meshgrid(*a), -1).reshape(-1, len(a))

    assert np.array_equal(cartesian_product(range(3), range(2)), golden["cart_3_2"])
    assert np.array_equal(cartesian_product((0, 180), (0, 120, 240), (0, 90, 180, 270), (0, 60, 120, 180, 240, 300)),
                          golden["cart_angles"])
    assert np.array_equal(cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4), golden["cart_6x4"])
    rng = np.random.default_rng(4)
    cases = [
        [(0, 60, 120, 180, 240, 300)] * 6,                                   # 46 656 x 6, the library's path
        [np.arange(5000)], [np.arange(300), np.arange(50)],                   # one and two arrays
        [np.linspace(0, 1, 70), np.arange(90), (1.5, 2.5, 3.5)],              # mixed -> float64
        [np.arange(10, dtype=np.int32), np.arange(1000, dtype=np.int16), (True, False)],  # -> int32
        [np.arange(70, dtype=np.uint8), np.arange(90, dtype=np.float32)],     # -> float32
        [np.arange(100), np.array([], dtype=np.int64)],                       # an empty factor
        [np.arange(6).reshape(2, 3), np.arange(2000)],                        # N-D input is flattened
        [rng.integers(-5, 5, 17), rng.integers(0, 9, 23), rng.integers(0, 3, 19)],
        [np.array(["a", "b"]), np.array(["x", "y", "z"])],                    # strings: NumPy path
        [np.arange(3000, dtype=np.uint64), np.arange(3)],                     # uint64: NumPy path (no int64 detour)
    ]
    from firecode_amd.utils import cartesian_rows_at

    for arrays in cases:
        mine, want = cartesian_product(*arrays), ref(*arrays)
        assert mine.shape == want.shape and mine.dtype == want.dtype and np.array_equal(mine, want)
        if len(want):  # rows of the product from their row numbers alone (what the device-generated grid leaves the host)
            pick = rng.integers(0, len(want), 200)
            got = cartesian_rows_at(arrays, pick)
            assert got.shape == (200, want.shape[1]) and np.array_equal(got, want[pick])
    assert np.array_equal(cartesian_rows_at(cases[0], np.arange(6 ** 6)), ref(*cases[0]))


def test_ensemble_energy_threshold_against_the_reference_vectors_and_the_literal_loop(golden):
    """Ensemble.dynamic_energy_thr / energy_pruning (firecode/ensemble.py:117-169) without the reference's loop:
    the vectors its own methods produced, and the literal loop on energies in RANDOM order (the rule returns
    the first qualifying energy in array order, not the smallest), with ties and with nothing qualifying"""
    from firecode_amd import refining
    from firecode_amd.ensemble import Ensemble
    from oracle import cpu_ref as o

    en = np.asarray(golden["enp_energies"])
    e = Ensemble(atoms=np.array(["C"] * 7), coords=np.zeros((len(en), 7, 3)), energies=en.copy(), logfunction=None)
    assert e.dynamic_energy_thr(10.0, verbose=False) == float(golden["enp_thr10"])
    assert e.dynamic_energy_thr(0.5, verbose=False) == float(golden["enp_thr0p5"])
    e.energy_pruning(10.0, verbose=False)
    assert len(e.coords) == int(golden["enp_kept10"]) == len(e.energies)
    rng = np.random.default_rng(9)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        rel = np.round(rng.uniform(0, 30, size=n), 1 if trial % 2 else 6)   # one decimal: ties
        rel[int(rng.integers(0, n))] = 0.0
        thr, keep_min = float(rng.choice([0.05, 0.5, 3.0, 10.0, 40.0])), float(rng.choice([0.1, 0.5, 0.9]))
        want = o.dynamic_energy_thr(rel, thr, keep_min)
        assert refining.first_threshold_keeping(rel, thr, keep_min)[0] == want
        msgs, shifted = [], rel + 5.0
        want = o.dynamic_energy_thr(shifted - shifted.min(), thr, keep_min)
        ens = Ensemble(atoms=np.array(["C"]), coords=np.zeros((n, 1, 3)), energies=shifted, logfunction=msgs.append)
        assert ens.dynamic_energy_thr(thr, keep_min) == want and (len(msgs) == 1) == (want != thr)
        assert refining.dynamic_energy_thr(shifted, thr, keep_min) == want
