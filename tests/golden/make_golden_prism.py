"""Pinning kit for the THIRD-PARTY rows: write tests/golden/prism_v1.npz from the real
``prism_pruner`` package (the reference pins 0.0.7, pixi.lock:5054-5063).

    python tests/golden/make_golden_prism.py          # anywhere `import prism_pruner` works

The package is not vendored in the reference tree and not installed in the authoring image, so
this script could not be run there: the rows it covers -- rmsd_and_max, get_alignment_matrix,
prune_by_rmsd, prune_by_moment_of_inertia, prune_by_rmsd_rot_corr, align_structures,
rotate_dihedral, dihedral, rot_mat_from_pointer, vec_angle, normalize, get_inertia_moments,
graphize / get_sp_n / is_amide_n / is_ester_o / get_double_bonds_indices -- stay "parity unpinned"
until someone runs it once on a machine that has the package and commits the .npz.
``tests/test_prism_golden.py`` then checks the oracle (CPU) and the HIP path (GPU) against every
array in it, and reports which value of each ``CONVENTIONS`` switch reproduces the package's
masks.  Nothing of this repository is imported here except the seeded input generator; every
section is independent -- one that fails (an API that moved between versions) is recorded under
``errors`` and the others are still written.
"""

import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from firecode_amd import synthetic as syn  # noqa: E402  (input generation only)


def main():
    try:
        import prism_pruner
        from prism_pruner import algebra as pa
        from prism_pruner import graph_manipulations as pg
        from prism_pruner import pruner as pp
        from prism_pruner import rmsd as pr
        from prism_pruner import utils as pu
    except ImportError as exc:
        sys.exit(f"prism_pruner is not importable here ({exc}); nothing written")

    G, errors = {"version": np.array(getattr(prism_pruner, "__version__", "unknown"))}, []
    rng = np.random.default_rng(20261004)

    def section(name, fn):
        try:
            fn()
        except Exception:  # noqa: BLE001
            errors.append(f"[{name}]\n{traceback.format_exc()}")

    # ---- inputs: seeded synthetic ensembles (clustered, with hydrogens) + energies ----------
    X, _, asg = syn.synthetic_ensemble(240, 18, seed=71)
    atoms = np.array((["C", "C", "H", "N", "O", "H"] * 3))
    en = rng.uniform(0.0, 3.0, size=len(X))
    Xc = syn.continuous_ensemble(160, 18, seed=5, thr=0.5)
    G.update(X=X, atoms=atoms, energies=en, Xc=Xc)

    def s_rmsd():
        iu, ju = np.triu_indices(40, 1)
        out = np.array([pr.rmsd_and_max(X[a], X[b], center=True) for a, b in zip(iu, ju)])
        G["rm_pairs"], G["rm_center_true"] = np.stack([iu, ju], 1), out
        Xz = X - X.mean(axis=1, keepdims=True)
        G["rm_center_false_on_centred"] = np.array([pr.rmsd_and_max(Xz[a], Xz[b]) for a, b in zip(iu, ju)])
        G["rm_center_false_on_raw"] = np.array([pr.rmsd_and_max(X[a], X[b]) for a, b in zip(iu[:60], ju[:60])])
        G["alignment_matrix"] = np.array([pr.get_alignment_matrix(Xz[a], Xz[b]) for a, b in zip(iu[:60], ju[:60])])

    def s_prune_rmsd():
        for name, Y in (("clustered", X), ("continuous", Xc)):
            for thr in (0.25, 0.5, 1.0):
                G[f"prune_rmsd_{name}_{thr}"] = np.asarray(pp.prune_by_rmsd(Y, atoms, thr)[1])
            G[f"prune_rmsd_{name}_default"] = np.asarray(pp.prune_by_rmsd(Y, atoms)[1])  # ensemble.py:230 passes no threshold
        G["prune_rmsd_energies_1.0"] = np.asarray(pp.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0)[1])
        G["prune_rmsd_energies_0.2"] = np.asarray(pp.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=0.2)[1])
        # thresholds exactly ON the rmsd of a pair decide "<" against "<="
        Xz = X[:, atoms != "H"]
        Xz = Xz - Xz.mean(axis=1, keepdims=True)
        r01 = float(pr.rmsd_and_max(Xz[0], Xz[1])[0])
        G["prune_rmsd_tie_thr"] = np.array(r01)
        G["prune_rmsd_tie_pair"] = np.asarray(pp.prune_by_rmsd(X[:2], atoms, r01)[1])
        # which member of a similar pair falls: two near-identical structures, then three in a chain
        twin = np.stack([X[0], X[0] + 1e-3])
        G["prune_rmsd_twin"] = np.asarray(pp.prune_by_rmsd(twin, atoms, 0.5)[1])
        G["prune_rmsd_twin_energies"] = np.asarray(pp.prune_by_rmsd(twin, atoms, 0.5, energies=np.array([1.0, 0.0]), max_dE=5.0)[1])
        # max deviation rule: one atom displaced far, rmsd still below threshold
        spike = np.stack([X[0], X[0].copy()])
        spike[1, 0] += np.array([0.0, 0.0, 1.3])
        G["prune_rmsd_spike_in"] = spike
        for thr in (0.4, 0.5, 0.6, 0.7, 0.8, 1.0, 1.4):
            G[f"prune_rmsd_spike_{thr}"] = np.asarray(pp.prune_by_rmsd(spike, atoms, thr)[1])

    def s_prune_moi():
        G["moi_moments"] = np.array([pa.get_inertia_moments(x, np.ones(len(atoms))) for x in X[:40]])
        G["prune_moi"] = np.asarray(pp.prune_by_moment_of_inertia(X, atoms)[1])
        G["prune_moi_continuous"] = np.asarray(pp.prune_by_moment_of_inertia(Xc, atoms)[1])
        G["prune_moi_energies"] = np.asarray(pp.prune_by_moment_of_inertia(X, atoms, energies=en, max_dE=1.0)[1])
        # the tolerance: scale one structure isotropically by 1 +- eps and see where similarity ends
        for eps in (0.002, 0.004, 0.006, 0.01, 0.02):
            pair = np.stack([X[0] - X[0].mean(0), (X[0] - X[0].mean(0)) * (1.0 + eps)])
            G[f"prune_moi_scaled_{eps}"] = np.asarray(pp.prune_by_moment_of_inertia(pair, atoms)[1])

    def s_algebra():
        P = rng.normal(size=(50, 4, 3))
        G["dihedral_in"] = P
        G["dihedral_out"] = np.array([pa.dihedral(p) for p in P])
        ax = rng.normal(size=(30, 3))
        ang = rng.uniform(-360, 360, size=30)
        G["rmfp_axis"], G["rmfp_angle"] = ax, ang
        G["rmfp_out"] = np.array([pa.rot_mat_from_pointer(a, t) for a, t in zip(ax, ang)])
        v = rng.normal(size=(30, 2, 3))
        G["vec_angle_in"] = v
        G["vec_angle_out"] = np.array([pa.vec_angle(a, b) for a, b in v])
        G["normalize_out"] = np.array([pa.normalize(a) for a in v[:, 0]])

    def s_utils():
        base = syn.synthetic_skeleton(20, np.random.default_rng(3))
        mask = np.zeros(20, dtype=bool)
        mask[9:] = True
        tors = (6, 7, 8, 9)
        G["rd_base"], G["rd_mask"], G["rd_torsion"] = base, mask, np.array(tors)
        G["rd_angles"] = np.array([5.0, -5.0, 60.0, 120.0, 180.0, 270.0])
        G["rd_out"] = np.array([pu.rotate_dihedral(base.copy(), tors, a, mask=mask) for a in G["rd_angles"]])
        G["rd_input_after"] = base  # tells whether rotate_dihedral works in place
        G["align_structures"] = np.asarray(pu.align_structures(X[:30].copy()))
        G["align_structures_idx"] = np.asarray(pu.align_structures(X[:30].copy(), np.array([0, 1, 3, 4, 6])))

    def s_graph():
        # the reference's own fixture molecule (butane) as data: bonds, hybridisation, double bonds
        import io

        txt = open(os.path.join(HERE, "butane_fixture.xyz")).read() if os.path.exists(os.path.join(HERE, "butane_fixture.xyz")) else None
        if txt is None:
            g0 = np.load(os.path.join(HERE, "intree_v1.npz"))
            txt = str(g0["fx_butane_text"])
        lines = txt.splitlines()
        n = int(lines[0])
        a = np.array([ln.split()[0] for ln in lines[2: 2 + n]])
        c = np.array([[float(x) for x in ln.split()[1:4]] for ln in lines[2: 2 + n]])
        g = pg.graphize(a, c)
        G["graph_butane_edges"] = np.array(sorted(tuple(sorted(e)) for e in g.edges), dtype=np.int64)
        G["graph_butane_sp_n"] = np.array([-1 if pg.get_sp_n(i, g) is None else pg.get_sp_n(i, g) for i in range(n)])
        G["graph_butane_double_bonds"] = np.array(pu.get_double_bonds_indices(c, a), dtype=np.int64).reshape(-1, 2)
        G["d_min_bond_CC_CH_default"] = np.array([pg.d_min_bond("C", "C"), pg.d_min_bond("C", "H")])
        _ = io

    def s_rot_corr():
        import networkx as nx

        # a chain whose end carries three equivalent carbons (tBu-like): rotamers by 120 degrees
        base = np.array([[-3.7, 1.4, 0.3], [-2.2, 1.3, 0.0], [-1.5, 0.0, 0.0], [0.0, 0.0, 0.0],
                         [0.50, 1.39, 0.25], [0.50, -0.91, 1.08], [0.50, -0.48, -1.33]])
        at = np.array(["C"] * 7)
        g = pg.graphize(at, base)
        mask = np.zeros(7, dtype=bool)
        mask[4:] = True
        confs = np.array([pu.rotate_dihedral(base.copy(), (1, 2, 3, 4), a, mask=mask) for a in (0.0, 120.0, 240.0, 60.0)])
        G["rotcorr_in"], G["rotcorr_edges"] = confs, np.array(sorted(g.edges), dtype=np.int64)
        G["rotcorr_mask"] = np.asarray(pp.prune_by_rmsd_rot_corr(confs, at, g, max_rmsd=0.25)[1])
        G["rotcorr_plain_mask"] = np.asarray(pp.prune_by_rmsd(confs, at, 0.25)[1])
        _ = nx

    for name, fn in (("rmsd", s_rmsd), ("prune_rmsd", s_prune_rmsd), ("prune_moi", s_prune_moi), ("algebra", s_algebra),
                     ("utils", s_utils), ("graph", s_graph), ("rot_corr", s_rot_corr)):
        section(name, fn)
    G["errors"] = np.array("\n".join(errors))
    out = os.path.join(HERE, "prism_v1.npz")
    np.savez_compressed(out, **G)
    print(f"wrote {out}: {len(G)} arrays, {len(errors)} failed section(s)")
    if errors:
        print("\n".join(errors))


if __name__ == "__main__":
    main()
