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
