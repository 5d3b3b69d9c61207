"""Generate tests/golden/intree_v1.npz from the REFERENCE'S OWN function objects.

Run in the authoring container only (the reference never travels):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import recipe: the reference needs Python >= 3.12
(``typing.Self``) and the un-vendored third-party ``prism_pruner``; both are
satisfied with in-memory placeholders *that raise if called*, so only
functions whose body is NumPy/SciPy/networkx run.  Every vector below is
therefore the output of reference code alone -- nothing of this repository is
imported here.  Inputs are seeded; the committed .npz holds inputs + outputs.
"""

import io
import os
import sys
import tempfile
import types
import typing

import numpy as np
import typing_extensions

typing.Self = typing_extensions.Self
for _name in ("prism_pruner", "prism_pruner.algebra", "prism_pruner.graph_manipulations",
              "prism_pruner.pruner", "prism_pruner.utils", "prism_pruner.rmsd",
              "prism_pruner.periodic_table"):
    _m = types.ModuleType(_name)

    def _ga(k, _n=_name):
        def _raise(*a, **kw):
            raise RuntimeError(f"placeholder {_n}.{k} called: not reference code")
        return _raise

    _m.__getattr__ = _ga
    sys.modules[_name] = _m

import firecode.algebra as fa  # noqa: E402
import firecode.embeds as fe  # noqa: E402
import firecode.ensemble as fens  # noqa: E402
import firecode.torsion_module as ft  # noqa: E402
import firecode.utils as fu  # noqa: E402

rng = np.random.default_rng(20260821)
G = {}

# ---- align_vec_pair (algebra.py:28-49) --------------------------------------
ref = rng.normal(size=(64, 2, 3))
tgt = rng.normal(size=(64, 2, 3))
tgt[:8] = ref[:8] @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])  # exact rotations
tgt[8:12, 1] = tgt[8:12, 0] * 2.0  # collinear targets (rank-1 covariance)
G["avp_ref"], G["avp_tgt"] = ref, tgt
G["avp_out"] = np.array([fa.align_vec_pair(r, t) for r, t in zip(ref, tgt)])

# ---- count_clashes (algebra.py:52-54) ---------------------------------------
cc_in = rng.normal(scale=1.2, size=(40, 30, 3))
cc_in[0, 1] = cc_in[0, 0]            # coincident atoms: excluded by > 0
cc_in[1, 1] = cc_in[1, 0] + [0.5, 0, 0]   # exactly on the threshold
G["cc_in"] = cc_in
G["cc_out"] = np.array([fa.count_clashes(c) for c in cc_in], dtype=np.int64)

# ---- compenetration_check (utils.py:507-575) --------------------------------
cp_in = rng.normal(scale=2.0, size=(60, 36, 3))
G["cp_in"] = cp_in
G["cp_none"] = np.array([fu.compenetration_check(c) for c in cp_in])
G["cp_none_mc2"] = np.array([fu.compenetration_check(c, max_clashes=2) for c in cp_in])
for thr in (1.0, 1.5):
    for mc in (0, 3):
        G[f"cp_bi_{thr}_{mc}"] = np.array(
            [fu.compenetration_check(c, ids=[20, 16], thresh=thr, max_clashes=mc) for c in cp_in])
        G[f"cp_tri_{thr}_{mc}"] = np.array(
            [fu.compenetration_check(c, ids=[12, 14, 10], thresh=thr, max_clashes=mc) for c in cp_in])
# graph mode, with a chain graph
import networkx as nx  # noqa: E402

chain = nx.path_graph(36)
G["cp_graph_edges"] = np.array(chain.edges, dtype=np.int64)
cpg_in = rng.normal(scale=3.0, size=(60, 36, 3))
G["cpg_in"] = cpg_in
for mc in (0, 2):
    G[f"cp_graph_{mc}"] = np.array(
        [fu.compenetration_check(c, graph=chain, thresh=1.2, max_clashes=mc) for c in cpg_in])

# ---- cartesian_product (utils.py:219-221) -----------------------------------
G["cart_3_2"] = fu.cartesian_product(range(3), range(2))
G["cart_angles"] = fu.cartesian_product((0, 180), (0, 120, 240), (0, 90, 180, 270), (0, 60, 120, 180, 240, 300))
G["cart_6x4"] = fu.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4)

# ---- rotation_matrix_from_vectors (non-antiparallel) (utils.py:224-249) -----
v1 = rng.normal(size=(32, 3))
v2 = rng.normal(size=(32, 3))
v2[0] = v1[0] * 3.0  # parallel -> identity branch
G["rmv_v1"], G["rmv_v2"] = v1, v2
G["rmv_out"] = np.array([fu.rotation_matrix_from_vectors(a, b) for a, b in zip(v1, v2)])

# ---- polygonize (utils.py:252-312) ------------------------------------------
G["poly2_in"] = np.array([2.3, 3.1])
G["poly2_out"] = fu.polygonize(G["poly2_in"])
G["poly3_in"] = np.array([2.0, 2.5, 3.0])
G["poly3_out"] = fu.polygonize(G["poly3_in"])


# ---- get_embed (embeds.py:808-817) ------------------------------------------
class _Mol:
    pass


def _rot(r):
    q, rr = np.linalg.qr(r.normal(size=(3, 3)))
    q = q * np.sign(np.diag(rr))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


m1, m2 = _Mol(), _Mol()
m1.coords = rng.normal(scale=2.0, size=(4, 17, 3))
m2.coords = rng.normal(scale=2.0, size=(3, 23, 3))
ge_R = np.array([[_rot(rng), _rot(rng)] for _ in range(12)])
ge_t = rng.normal(scale=4.0, size=(12, 2, 3))
ge_ids = np.stack([rng.integers(0, 4, size=12), rng.integers(0, 3, size=12)], axis=1)
ge_out = []
for R, t, ids in zip(ge_R, ge_t, ge_ids):
    m1.rotation, m1.position = R[0], t[0]
    m2.rotation, m2.position = R[1], t[1]
    ge_out.append(fe.get_embed([m1, m2], ids))
G["ge_c1"], G["ge_c2"], G["ge_R"], G["ge_t"], G["ge_ids"] = m1.coords, m2.coords, ge_R, ge_t, ge_ids
G["ge_out"] = np.array(ge_out)

# ---- _get_cyclical_reactive_indices (embeds.py:753-784), both branches ------------------------
def _piv(a, b):
    return types.SimpleNamespace(start_atom=types.SimpleNamespace(cumnum=a), end_atom=types.SimpleNamespace(cumnum=b))


cri_cum2 = np.array([[3, 9], [20, 25]])
cri_cum3 = np.array([[2, 7], [13, 19], [31, 25]])
emb2 = types.SimpleNamespace(objects=[None, None])
emb3 = types.SimpleNamespace(objects=[None, None, None])
G["cri_cum2"], G["cri_cum3"] = cri_cum2, cri_cum3
G["cri_out2"] = np.array([fe._get_cyclical_reactive_indices(emb2, [_piv(*c) for c in cri_cum2.tolist()], n) for n in range(2)])
G["cri_out3"] = np.array([fe._get_cyclical_reactive_indices(emb3, [_piv(*c) for c in cri_cum3.tolist()], n) for n in range(8)])

# ---- torsion_comp_check (torsion_module.py:894-918) -------------------------
tc_in = rng.normal(scale=1.8, size=(50, 40, 3))
tc_mask = rng.random(size=(50, 40)) < 0.4
tc_tors = np.array([rng.choice(40, size=4, replace=False) for _ in range(50)], dtype=np.int64)
G["tc_in"], G["tc_mask"], G["tc_tors"] = tc_in, tc_mask, tc_tors
G["tc_out"] = np.array([ft.torsion_comp_check(c, tuple(t), m.copy(), thresh=1.5)
                        for c, t, m in zip(tc_in, tc_tors, tc_mask)])
G["tc_out_mc2"] = np.array([ft.torsion_comp_check(c, tuple(t), m.copy(), thresh=1.5, max_clashes=2)
                            for c, t, m in zip(tc_in, tc_tors, tc_mask)])

# ---- _get_rotation_mask (torsion_module.py:354-382) --------------------------
rm_edges = [(0, 1), (1, 2), (2, 3), (3, 4), (3, 5), (3, 6), (0, 7), (0, 8), (4, 9), (5, 10), (6, 11), (8, 12), (12, 13)]
rm_graph = nx.Graph(rm_edges)
rm_tors = np.array([(1, 2, 3, 4), (2, 1, 0, 7), (0, 1, 2, 3), (3, 2, 1, 0), (1, 0, 8, 12), (7, 0, 8, 12)], dtype=np.int64)
G["rotmask_edges"] = np.array(rm_edges, dtype=np.int64)
G["rotmask_torsions"] = rm_tors
G["rotmask_out"] = np.array([ft._get_rotation_mask(rm_graph, tuple(int(x) for x in t)) for t in rm_tors])

# ---- tfd_similarity + prune_conformers_tfd loop (torsion_module.py:957-1067) -
tf_a = rng.uniform(-180, 180, size=(40, 6))
tf_b = tf_a + rng.normal(scale=2.0, size=tf_a.shape)
tf_b[::3] = rng.uniform(-180, 180, size=tf_b[::3].shape)
tf_b[1] = tf_a[1] + [359.0, 0, 0, 0, 0, 0]  # wrap-around
G["tfd_a"], G["tfd_b"] = tf_a, tf_b
G["tfd_out"] = np.array([ft.tfd_similarity(a, b, thresh=10) for a, b in zip(tf_a, tf_b)])
# long fingerprints: np.sum switches to its 8-lane pairwise order at Q >= 8
tfl_a = rng.uniform(-180, 180, size=(400, 17))
tfl_b = tfl_a + rng.normal(scale=0.72, size=tfl_a.shape)   # sums straddle the threshold
G["tfdl_a"], G["tfdl_b"] = tfl_a, tfl_b
G["tfdl_out"] = np.array([ft.tfd_similarity(a, b, thresh=10) for a, b in zip(tfl_a, tfl_b)])


def _tfd_case(n, q, n_clusters, noise, seed):
    r = np.random.default_rng(seed)
    centres = r.choice([-120.0, 0.0, 60.0, 120.0, 180.0, -60.0], size=(n_clusters, q))
    asg = r.integers(0, n_clusters, size=n)
    return centres[asg] + r.normal(scale=noise, size=(n, q))


for name, (n, q, k, noise, seed) in {
    "tfdp_small": (60, 4, 12, 0.8, 1),
    "tfdp_mid": (333, 5, 40, 1.0, 2),
    "tfdp_big": (1500, 6, 200, 0.9, 3),
    "tfdp_dense": (400, 3, 6, 1.5, 4),
    "tfdp_q8": (500, 8, 60, 0.55, 5),
    "tfdp_q11": (300, 11, 40, 0.45, 6),
    "tfdp_q19": (200, 19, 25, 0.26, 7),
}.items():
    tf = _tfd_case(n, q, k, noise, seed)
    saved = ft._get_tf_mat
    ft._get_tf_mat = lambda structures, quadruplets, _tf=tf: _tf
    try:
        dummy = np.zeros((n, 4, 3))
        _, mask = ft.prune_conformers_tfd(dummy, np.zeros((q, 4), dtype=int), thresh=10)
    finally:
        ft._get_tf_mat = saved
    G[name + "_tf"], G[name + "_mask"] = tf, mask

# ---- Ensemble xyz format + energy pruning (ensemble.py:58-169, 284-297) ------
ens_atoms = np.array(["C", "H", "O", "N", "Cl", "H", "C"])
ens_coords = rng.normal(scale=3.0, size=(5, 7, 3))
ens = fens.Ensemble(atoms=ens_atoms, coords=ens_coords, basename="golden", logfunction=None)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "golden.xyz")
    ens.to_xyz(path)
    text = open(path).read()
    saved_pt = fens.pt
    fens.pt = types.SimpleNamespace(number=lambda s: 0)
    try:
        back = fens.Ensemble.from_xyz(path)
    finally:
        fens.pt = saved_pt
G["ens_atoms"], G["ens_coords"] = ens_atoms, ens_coords
G["ens_text"] = np.array(text)
G["ens_back_coords"] = back.coords
G["ens_back_atoms"] = back.atoms

# ---- the reference's own test fixtures through its own reader (ensemble.py:58-98) -----------
# data files of firecode/tests: text in, (atoms, coords) out; plus the in-tree clash functions
# on these real molecules
_fx = {"anti_to_gauche": "multithread_refine/anti_to_gauche.xyz", "catalyst": "operator_firecode_search/catalyst.xyz",
       "salt": "operator_crest_search/salt.xyz", "butane": "operator_rdkit_search/butane.xyz",
       "propane_ts": "propane_ts.xyz", "c2h4_hyper": "embed_cyclical/C2H4_hypermolecule.xyz"}
_tests_dir = os.path.join(os.path.dirname(fens.__file__), "tests")
saved_pt = fens.pt
fens.pt = types.SimpleNamespace(number=lambda s: 0)
try:
    for name, rel in _fx.items():
        path = os.path.join(_tests_dir, rel)
        e = fens.Ensemble.from_xyz(path)
        G[f"fx_{name}_text"] = np.array(open(path).read())
        G[f"fx_{name}_atoms"] = np.array(e.atoms)
        G[f"fx_{name}_coords"] = np.array(e.coords)
        G[f"fx_{name}_clashes"] = np.array([fa.count_clashes(c) for c in e.coords], dtype=np.int64)
        A_ = e.coords.shape[1]
        G[f"fx_{name}_frag"] = np.array([[fu.compenetration_check(c, ids=[A_ // 2, A_ - A_ // 2], thresh=1.6, max_clashes=mc)
                                          for mc in (0, 1, 2, 4, 8)] for c in e.coords])
finally:
    fens.pt = saved_pt
G["fx_names"] = np.array(list(_fx))

en = np.sort(rng.uniform(0, 30, size=50))
en[0] = 0.0
ens2 = fens.Ensemble(atoms=ens_atoms, coords=rng.normal(size=(50, 7, 3)), energies=en.copy(), logfunction=None)
G["enp_energies"] = en
G["enp_thr10"] = np.float64(ens2.dynamic_energy_thr(10.0, verbose=False))
G["enp_thr0p5"] = np.float64(ens2.dynamic_energy_thr(0.5, verbose=False))
ens2.energy_pruning(10.0, verbose=False)
G["enp_kept10"] = np.int64(len(ens2.coords))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "intree_v1.npz")
np.savez_compressed(out, **G)
print("wrote", out, {k: np.asarray(v).shape for k, v in G.items()})
