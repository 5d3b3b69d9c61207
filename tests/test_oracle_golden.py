"""The oracle against the golden vectors produced by the reference's own
in-tree functions (tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from oracle import cpu_ref as o


def test_align_vec_pair(golden):
    out = np.array([o.align_vec_pair(r, t) for r, t in zip(golden["avp_ref"], golden["avp_tgt"])])
    assert np.array_equal(out, golden["avp_out"])  # same LAPACK call, same bits
    # the N-atom Kabsch uses the same convention
    out2 = np.array([o.get_alignment_matrix(r, t) for r, t in zip(golden["avp_ref"], golden["avp_tgt"])])
    ok = np.ones(len(out2), dtype=bool)
    ok[8:12] = False  # rank-1 covariance: the optimal rotation is not unique
    assert np.allclose(out2[ok], golden["avp_out"][ok], atol=1e-12)


def test_count_clashes(golden):
    out = np.array([o.count_clashes(c) for c in golden["cc_in"]])
    assert np.array_equal(out, golden["cc_out"])


def test_compenetration_check(golden):
    cp = golden["cp_in"]
    assert np.array_equal([o.compenetration_check(c) for c in cp], golden["cp_none"])
    assert np.array_equal([o.compenetration_check(c, max_clashes=2) for c in cp], golden["cp_none_mc2"])
    for thr in (1.0, 1.5):
        for mc in (0, 3):
            bi = [o.compenetration_check(c, ids=[20, 16], thresh=thr, max_clashes=mc) for c in cp]
            tri = [o.compenetration_check(c, ids=[12, 14, 10], thresh=thr, max_clashes=mc) for c in cp]
            assert np.array_equal(bi, golden[f"cp_bi_{thr}_{mc}"])
            assert np.array_equal(tri, golden[f"cp_tri_{thr}_{mc}"])
    edges = golden["cp_graph_edges"]
    for mc in (0, 2):
        out = [o.compenetration_check(c, graph_edges=edges, thresh=1.2, max_clashes=mc) for c in golden["cpg_in"]]
        assert np.array_equal(out, golden[f"cp_graph_{mc}"])


def test_cartesian_product(golden):
    assert np.array_equal(o.cartesian_product(range(3), range(2)), golden["cart_3_2"])
    assert np.array_equal(
        o.cartesian_product((0, 180), (0, 120, 240), (0, 90, 180, 270), (0, 60, 120, 180, 240, 300)),
        golden["cart_angles"])
    assert np.array_equal(o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4), golden["cart_6x4"])


def test_rotation_matrix_from_vectors(golden):
    out = np.array([o.rotation_matrix_from_vectors(a, b) for a, b in zip(golden["rmv_v1"], golden["rmv_v2"])])
    assert np.array_equal(out, golden["rmv_out"])


def test_polygonize(golden):
    assert np.array_equal(o.polygonize(golden["poly2_in"]), golden["poly2_out"])
    assert np.array_equal(o.polygonize(golden["poly3_in"]), golden["poly3_out"])


def test_get_embed(golden):
    for R, t, ids, exp in zip(golden["ge_R"], golden["ge_t"], golden["ge_ids"], golden["ge_out"]):
        out = o.get_embed([golden["ge_c1"][ids[0]], golden["ge_c2"][ids[1]]], R, t)
        assert np.array_equal(out, exp)


def test_torsion_comp_check(golden):
    out = [o.torsion_comp_check(c, tuple(t), m.copy()) for c, t, m in
           zip(golden["tc_in"], golden["tc_tors"], golden["tc_mask"])]
    assert np.array_equal(out, golden["tc_out"])
    out = [o.torsion_comp_check(c, tuple(t), m.copy(), max_clashes=2) for c, t, m in
           zip(golden["tc_in"], golden["tc_tors"], golden["tc_mask"])]
    assert np.array_equal(out, golden["tc_out_mc2"])


def test_tfd_similarity(golden):
    out = [o.tfd_similarity(a, b) for a, b in zip(golden["tfd_a"], golden["tfd_b"])]
    assert np.array_equal(out, golden["tfd_out"])


@pytest.mark.parametrize("name", ["tfdp_small", "tfdp_mid", "tfdp_big", "tfdp_dense"])
def test_prune_tfd_loop(golden, name):
    mask = o.prune_tfd_from_tf_mat(golden[name + "_tf"], thresh=10)
    assert np.array_equal(mask, golden[name + "_mask"])
    assert 0 < mask.sum() < len(mask)


def test_xyz_format(golden):
    text = o.ensemble_to_xyz_text(golden["ens_atoms"], golden["ens_coords"], basename="golden")
    assert text == str(golden["ens_text"])
    atoms, coords = o.ensemble_from_xyz_text(text)
    assert np.array_equal(coords, golden["ens_back_coords"])
    assert np.array_equal(atoms, golden["ens_back_atoms"])


def test_reference_fixture_files(golden, tmp_path):
    """the reference's own test data files (firecode/tests/**.xyz): its reader's output and its
    in-tree clash functions on those molecules"""
    for name in golden["fx_names"]:
        text = str(golden[f"fx_{name}_text"])
        atoms, coords = o.ensemble_from_xyz_text(text)
        assert np.array_equal(np.asarray(atoms), golden[f"fx_{name}_atoms"])
        assert np.array_equal(coords, golden[f"fx_{name}_coords"])
        A = coords.shape[1]
        assert np.array_equal([o.count_clashes(c) for c in coords], golden[f"fx_{name}_clashes"])
        frag = [[o.compenetration_check(c, ids=[A // 2, A - A // 2], thresh=1.6, max_clashes=mc) for mc in (0, 1, 2, 4, 8)]
                for c in coords]
        assert np.array_equal(frag, golden[f"fx_{name}_frag"])
