"""BASELINE.json's full-size configurations on the GPU, checked through
size-independent properties and random samples recomputed by the oracle
(the oracle cannot run these sizes whole)."""

import os

import numpy as np
import pytest

from firecode_amd import synthetic as syn
from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_cfg2_full_prune_properties(fc):
    """10 000 x 50, all-pairs RMSD prune at 0.5 A"""
    X, atoms, asg = syn.synthetic_ensemble(10000, 50, seed=2)
    pruned, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    K = len(np.unique(asg))
    assert mask.sum() == K and len(np.unique(asg[mask])) == K          # exactly one survivor per cluster
    # the survivor of a cluster is its LAST member (a structure falls to any later similar one)
    last = np.zeros(K, dtype=np.int64)
    last[asg] = np.arange(len(asg))
    assert np.array_equal(np.sort(np.flatnonzero(mask)), np.sort(last))
    # idempotence: nothing left to prune
    _, m2 = fc.pruner.prune_by_rmsd(pruned, atoms, 0.5)
    assert m2.all()
    # invariance under rigid motion and under reversing the order (mirror-image survivor set)
    rng = np.random.default_rng(0)
    Y = np.einsum("nij,naj->nai", np.array([syn.random_rotation(rng) for _ in range(len(X))]), X) + \
        rng.normal(scale=3.0, size=(len(X), 1, 3))
    _, m3 = fc.pruner.prune_by_rmsd(Y, atoms, 0.5)
    assert np.array_equal(m3, mask)
    # sampled pairs against the oracle's Kabsch
    iu = rng.integers(0, len(X), 3000)
    ju = rng.integers(0, len(X), 3000)
    r, d = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(r - r0).max() < TOL
    same = asg[iu] == asg[ju]
    assert np.abs(d - d0)[same].max() < TOL
    assert ((r < 0.5) == same).all()
    # similarity bits of a band of rows against the oracle decision
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, grey = ens.simbits(0.5, 1.0, row_begin=4000, row_end=4064)
    from firecode_amd._lib import unpack_bits

    S = unpack_bits(bits, len(X))
    expect = (asg[4000:4064, None] == asg[None, :]) & (np.arange(len(X))[None, :] > np.arange(4000, 4064)[:, None])
    assert np.array_equal(S, expect) and grey == 0


@pytest.mark.parametrize("n,a", [(7010, 224), (6000, 260), (4500, 320), (3000, 384)])
def test_large_compact_structures_prune_properties(fc, n, a):
    """The split-half screen's 32-column tile (193 ... 384 atoms; 7, 9, 10 and 12 k-steps here, the last three with row
    operands partly in scratch) at sizes with many row blocks and a partial last column tile, on globules with the
    radius of gyration of docked poses: one survivor per cluster, the LAST member of each, idempotent, and a band of
    rows of the similarity bits against the cluster structure."""
    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=800 + a, cluster_size=5, compact=True)
    pruned, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    from firecode_amd import _lib

    assert _lib.screen_last_kind() == 16
    K = len(np.unique(asg))
    assert mask.sum() == K and len(np.unique(asg[mask])) == K
    last = np.zeros(K, dtype=np.int64)
    last[asg] = np.arange(len(asg))
    assert np.array_equal(np.sort(np.flatnonzero(mask)), np.sort(last))
    _, m2 = fc.pruner.prune_by_rmsd(pruned, atoms, 0.5)
    assert m2.all()
    rng = np.random.default_rng(a)
    iu, ju = rng.integers(0, n, 2000), rng.integers(0, n, 2000)
    r, d = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(r - r0).max() < TOL and ((r < 0.5) == (asg[iu] == asg[ju])).all()
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, grey = ens.simbits(0.5, 1.0, row_begin=n - 200, row_end=n - 136)
    S = _lib.unpack_bits(bits, n)
    rows = np.arange(n - 200, n - 136)
    expect = (asg[rows, None] == asg[None, :]) & (np.arange(n)[None, :] > rows[:, None])
    assert np.array_equal(S, expect) and grey == 0


@pytest.mark.parametrize("n,a,seed", [(6010, 224, 2), (5000, 260, 2), (4000, 288, 2)])
def test_large_extended_structures_prune_properties(fc, n, a, seed):
    """The fp32 matrix-pipe screen's 32-column tile (214 ... 360 atoms, where the split-half bound's band is too wide:
    self-avoiding walks with a radius of gyration of 15-16 A) at sizes with many row blocks and a partial last column
    tile: the default selection takes it; one survivor per cluster, the LAST member of each, idempotent, sampled values
    against the oracle."""
    from firecode_amd import _lib

    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=seed, cluster_size=50)
    pruned, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert _lib.screen_last_kind() == 32
    K = len(np.unique(asg))
    assert mask.sum() == K and len(np.unique(asg[mask])) == K
    last = np.zeros(K, dtype=np.int64)
    last[asg] = np.arange(len(asg))
    assert np.array_equal(np.sort(np.flatnonzero(mask)), np.sort(last))
    _, m2 = fc.pruner.prune_by_rmsd(pruned, atoms, 0.5)
    assert m2.all()
    rng = np.random.default_rng(a)
    iu, ju = rng.integers(0, n, 1500), rng.integers(0, n, 1500)
    r, d = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(r - r0).max() < TOL and ((r < 0.5) == (asg[iu] == asg[ju])).all()


def test_cfg4_shape_80_atoms(fc):
    """the A = 80 kernel variant (one workgroup per CU) at a size the test can afford"""
    X, atoms, asg = syn.synthetic_ensemble(12000, 80, seed=4)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    K = len(np.unique(asg))
    assert mask.sum() == K and len(np.unique(asg[mask])) == K


def _headline_sample_pairs(n, rng, n_random=20000):
    """Pairs (i < j) that reach every structural corner of the complete-alignment launch: every row of the
    first and of the last row block (128 rows), rows on both sides of 16-, 64- and 128-row boundaries, columns in
    the last (partial) column tile and on both sides of 64-column boundaries, the region of the half-row-block
    items at the end of the item table (the last ~22 row blocks), and uniformly random pairs."""
    rows = set(range(0, min(128, n - 1))) | set(range(max(0, (n - 1) // 128 * 128 - 2), n - 1))
    for b in (16, 32, 48, 64, 112, 128, 256, 1024, (n // 256) * 128, (n // 128) * 128 - 128):
        rows |= {r for r in (b - 2, b - 1, b, b + 1) if 0 <= r < n - 1}
    for b in rng.integers(1, n // 128, 12):
        rows |= {r for r in (int(b) * 128 - 1, int(b) * 128, int(b) * 128 + 15, int(b) * 128 + 16, int(b) * 128 + 63,
                             int(b) * 128 + 64) if 0 <= r < n - 1}
    iu, ju = [], []
    last_tile = (n - 1) // 64 * 64
    for i in sorted(rows):
        cols = {i + 1, min(i + 2, n - 1), min(i + 3, n - 1), n - 1, n - 2}
        cols |= {c for c in ((i // 64 + 1) * 64 - 1, (i // 64 + 1) * 64, (i // 16 + 1) * 16 - 1, (i // 16 + 1) * 16,
                             last_tile - 1, last_tile, last_tile + 1) if i < c < n}
        cols |= {int(c) for c in rng.integers(i + 1, n, 24)}
        cols |= {int(c) for c in rng.integers(max(i + 1, last_tile), n, 4)}
        for c in sorted(cols):
            if c > i:
                iu.append(i)
                ju.append(c)
    a = rng.integers(0, n, n_random)
    b = rng.integers(0, n, n_random)
    keep = a != b
    iu += list(np.minimum(a, b)[keep])
    ju += list(np.maximum(a, b)[keep])
    # the tip of the triangle: the last 2 x 256 items of the launch are half-row-block items there
    a = rng.integers(max(0, n - 3000), n, 4000)
    b = rng.integers(max(0, n - 3000), n, 4000)
    keep = a != b
    iu += list(np.minimum(a, b)[keep])
    ju += list(np.maximum(a, b)[keep])
    return np.array(iu, dtype=np.int64), np.array(ju, dtype=np.int64)


def _check_against_oracle(X, iu, ju, r, d):
    """rmsd at 1e-10; max deviation at 1e-10 + the pair's own conditioning bound (oracle.rotation_error_bound_batch)"""
    err_r, err_d, slack = 0.0, 0.0, 0.0
    for k in range(0, len(iu), 8192):
        sl = slice(k, k + 8192)
        r0, d0 = o.rmsd_and_max_batch(X[iu[sl]], X[ju[sl]], center=True)
        bound = o.rotation_error_bound_batch(X[iu[sl]], X[ju[sl]], center=True)
        assert np.all(np.isfinite(bound))
        assert np.abs(r[sl] - r0).max() < TOL
        assert np.all(np.abs(d[sl] - d0) <= TOL + bound)
        err_r = max(err_r, float(np.abs(r[sl] - r0).max()))
        err_d = max(err_d, float(np.abs(d[sl] - d0).max()))
        slack = max(slack, float(bound.max()))
    return err_r, err_d, slack


@pytest.mark.parametrize("n,n_atoms,seed", [(10000, 50, 2), (12010, 80, 4), (3050, 160, 6), (2570, 260, 7)])
def test_complete_alignments_full_size_vs_oracle(fc, n, n_atoms, seed):
    """The bench's `value` kernel at the sizes it is timed at, both variants: k_simbits_screen_mfma<4, 2>
    (BASELINE configs[1]: 10 000 x 50, two workgroups per CU, half-row-block items at the end of the item table,
    16 columns in the last column tile) and <8, 2> (A = 80: one workgroup per CU; 12 010 conformers: 58 columns in
    the last tile, 106 rows in the last row block); and the narrow column tiles of larger structures: <8, 2, 32> at 160
    atoms (105 ... 208: 32 columns), <8, 2, 16> at 260 atoms (209 ... 416: 16 columns, k-steps in pairs, an odd number
    of them).  Both (N, N) outputs are downloaded whole: > 28 000 sampled
    pairs against the oracle's rmsd_and_max (firecode/utils.py:494-504), symmetry, exact-zero diagonal, and
    nothing outside what the kernel was asked for."""
    # (large structures: few cluster centres -- the generator redraws a centre until none of its atoms clash)
    X, atoms, asg = syn.synthetic_ensemble(n, n_atoms, seed=seed, cluster_size=5 if n_atoms <= 100 else 50)
    rng = np.random.default_rng(100 + n_atoms)
    iu, ju = _headline_sample_pairs(n, rng)
    assert len(iu) > 28000
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D, ms = ens.rmsd_and_max_all()
        # the bench's entry point (K passes, outputs resident) leaves the same numbers in the same places
        _, _, st, r_b, d_b = ens.bench_rmsd_and_max_all_sampled(iu, ju, reps=2)
    assert ms > 0 and int(st[0]) == n * (n - 1) // 2 and int(st[1]) < 1000 and int(st[2]) == 1  # st[1]: pairs redone by the Jacobi fix-up
    assert np.all(np.diag(R) == 0) and np.all(np.diag(D) == 0)
    for k in range(0, n, 2000):  # symmetric (the host mirrors the upper triangle), finite, non-negative
        blk = R[k:k + 2000]
        assert np.array_equal(blk, R[:, k:k + 2000].T) and np.array_equal(D[k:k + 2000], D[:, k:k + 2000].T)
        assert np.isfinite(blk).all() and (blk >= 0).all() and np.isfinite(D[k:k + 2000]).all()
        off = blk[np.arange(blk.shape[0])[:, None] + k != np.arange(n)[None, :]]
        assert off.min() > 0  # no element left unwritten (a pooled buffer is not zeroed): distinct conformers differ
        assert (D[k:k + 2000] >= blk * (1 - 1e-12)).all()  # the largest deviation is never below the rms one
    r, d = R[iu, ju], D[iu, ju]
    assert np.array_equal(r, r_b) and np.array_equal(d, d_b)
    _check_against_oracle(X, iu, ju, r, d)
    # the similarity decision from the two matrices reproduces the cluster structure of the whole ensemble
    same = asg[:, None] == asg[None, :]
    for k in range(0, n, 2000):
        assert np.array_equal((R[k:k + 2000] < 0.5) & (D[k:k + 2000] < 1.0), same[k:k + 2000])


@pytest.mark.parametrize("n,n_atoms,seed,worlds", [(10000, 50, 2, (2, 3, 8)), (6000, 80, 4, (2, 3)), (2100, 160, 6, (2, 3)), (1700, 260, 7, (3,))])
def test_complete_alignments_logical_ranks_equal_single_gpu(fc, n, n_atoms, seed, worlds):
    """What `bench.py --gpus N` measures: rank r computes the rows of the row blocks dealt to it in snake order
    (launch_rmsd_values(..., rank, world)).  Played on one GPU through fc_debug_comm_loopback, every rank's
    values on the rows it owns equal the single-GPU pass BIT FOR BIT, the owned pair counts add up, and the
    single-GPU pass equals the oracle on the same samples."""
    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    X, atoms, _ = syn.synthetic_ensemble(n, n_atoms, seed=seed, cluster_size=5 if n_atoms <= 100 else 50)
    rng = np.random.default_rng(7 + n_atoms)
    iu, ju = _headline_sample_pairs(n, rng, n_random=12000)
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            _, _, st1, r1, d1 = ens.bench_rmsd_and_max_all_sampled(iu, ju, reps=1)
            assert int(st1[0]) == n * (n - 1) // 2
            _check_against_oracle(X, iu, ju, r1, d1)
            for world in worlds:
                owner = fdist.owner_of_rows(n, world, 128)
                owned_total, seen = 0, np.zeros(len(iu), dtype=bool)
                for rk in range(world):
                    _lib.call("fc_debug_comm_loopback", rk, world)
                    mine = owner[iu] == rk
                    _, _, st, r, d = ens.bench_rmsd_and_max_all_sampled(iu[mine], ju[mine], reps=1)
                    assert np.array_equal(r, r1[mine]) and np.array_equal(d, d1[mine])
                    assert int(st[0]) == int((n - 1 - np.flatnonzero(owner == rk)).sum()) and int(st[1]) < 1000
                    owned_total += int(st[0])
                    seen |= mine
                assert owned_total == n * (n - 1) // 2 and seen.all()
    finally:
        _lib.call("fc_debug_comm_loopback", -1, 0)


def test_cfg5_full_pose_grid_samples(fc):
    """500 x 500 conformer pairs x 2 x 16 x 16 poses; 1 500 random poses recomputed"""
    n, A = 500, 40
    def mol(seed):
        X, _, _ = syn.synthetic_ensemble(n, A, seed=seed, cluster_size=1, sigma_cluster=0.25)
        X = X - X.reshape(-1, 3).mean(axis=0)
        return X, np.array([3, 7]), np.stack([X[:, 3] * 1.5, X[:, 7] * 1.5], axis=1)
    m1, r1, pv1 = mol(51)
    m2, r2, pv2 = mol(52)
    angles = np.arange(16) * 2 * 45.0 / 15 - 45.0
    ok, ms = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=1.5, max_clashes=0)
    assert ok.shape == (n, n, 2, 16, 16) and 0 < ok.sum() < ok.size
    rng = np.random.default_rng(5)
    for _ in range(1500):
        c2, c1, ori, i2, i1 = (rng.integers(0, s) for s in ok.shape)
        Ra, ta, Rb, tb = o.bimol_pose_transforms(m1[c1], m2[c2], r1, r2, pv1[c1], pv2[c2], (angles[i1], angles[i2]), ori)
        pose = o.get_embed([m1[c1], m2[c2]], [Ra, Rb], [ta, tb])
        assert ok[c2, c1, ori, i2, i1] == o.compenetration_check(pose, ids=[A, A], thresh=1.5)


def test_cfg3_full_scan_samples_and_tfd(fc):
    """8 torsions x 6-fold = 1 679 616 angle-sets; sampled against the oracle scan"""
    rng = np.random.default_rng(3)
    A, T = 50, 8
    base = syn.synthetic_skeleton(A, rng)
    centres = np.linspace(3, A - 6, T).astype(int)
    torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
    masks = np.zeros((T, A), dtype=bool)
    for t, c in enumerate(centres):
        masks[t, c + 2:] = True
    angles = fc.utils.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * T)
    assert angles.shape == (1679616, 8)
    out, rot = fc.torsion_module.torsion_scan(base, torsions, masks, angles, thresh=1.5)
    pick = rng.integers(0, len(angles), 250)
    ref_c, ref_r = o.torsion_scan(base, torsions, masks, angles[pick], thresh=1.5)
    assert np.array_equal(rot[pick], ref_r)
    assert np.abs(out[pick] - ref_c).max() < TOL
    keep = np.concatenate([[0], 1 + np.flatnonzero(rot != 0)])
    new = np.concatenate([base[None], out])[keep]
    del out
    tf = fc.torsion_module.get_tf_mat(new, torsions)
    assert np.abs(tf[pick] - o.get_tf_mat(new[pick], torsions)).max() < TOL
    mask = fc.torsion_module.prune_tfd_from_tf_mat(tf, 10)
    assert 0 < mask.sum() < len(mask)
    # first match at this size: bounded look-ahead + column chunks with window boxes (the default for long arrays)
    # == the one-phase kernel
    from firecode_amd import _lib as L

    fms = {}
    for look in ("0", None):
        if look is None:
            os.environ.pop("FC_TFD_LOOKAHEAD", None)
        else:
            os.environ["FC_TFD_LOOKAHEAD"] = look
        fms[look] = np.zeros(len(tf), dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(np.ascontiguousarray(tf)), len(tf), tf.shape[1], 10.0, L.pi(fms[look]))
    os.environ.pop("FC_TFD_LOOKAHEAD", None)
    assert np.array_equal(fms["0"], fms[None]) and (fms[None] < 0).sum() > 1000
    # the fused call of the csearch driver (fingerprints stay on the device) gives the same mask on the same rows
    rot_f, keep_f = fc.torsion_module.torsion_scan_tfd(base, torsions, masks, angles, torsions, thresh=1.5, tfd_thresh=10)
    assert np.array_equal(rot_f, rot)
    expect = np.zeros(len(angles) + 1, dtype=bool)
    expect[keep] = mask
    assert np.array_equal(keep_f, expect)
    # (the reference's TFD pruning is not idempotent -- first-match graph, masked structures
    # keep participating -- so the checks are: a literal-oracle run on a slice, and)
    sl = tf[100000:100700]
    assert np.array_equal(fc.torsion_module.prune_tfd_from_tf_mat(sl, 10), o.prune_tfd_from_tf_mat(sl, 10))
    # every removed structure has a TFD-similar structure somewhere (sampled)
    removed = np.flatnonzero(~mask)
    for i in rng.choice(removed, 15, replace=False):
        d = np.abs(tf - tf[i])
        d = np.abs(d - (d > 180) * 360).sum(axis=1)
        d[i] = 1e9
        assert d.min() < 10


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_multi_gpu_sizes_with_logical_ranks(fc, world):
    """the exact problem of `bench.py --gpus N` (10^4 sqrt(N) conformers, exchange capacity, item
    tables, half-item tail) with the N ranks played one after the other on ONE GPU: every
    rank's message fits, the replayed ladder gives the single-GPU mask, one survivor per cluster"""
    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    n = int(round(10000 * np.sqrt(world)))
    X, atoms, asg = syn.synthetic_ensemble(n, 50, seed=2)
    cap = fdist.exchange_cap(n, world)
    counts = []
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            ref, stats0 = ens.prune(0.5, 1.0)
            # fc_debug_comm_loopback: the library acts as rank r of `world`; its all-gather fills only slot r of
            # the receive buffer, so after the last rank the buffer holds every message (C path, no torch)
            for r in list(range(1, world)) + [0]:
                _lib.call("fc_debug_comm_loopback", r, world)
                mask, stats = ens.prune_sharded(0.5, 1.0)
                counts.append(int(stats[2]))
    finally:
        _lib.call("fc_debug_comm_loopback", -1, 0)
    counts = np.array(counts)
    assert np.array_equal(mask, ref)
    assert mask.sum() == len(np.unique(asg))
    assert int(counts.sum()) == stats0[2] and int(counts.max()) <= cap
    assert counts.max() - counts.min() < 0.1 * counts.mean()  # the snake deal balances the pairs too


@pytest.mark.parametrize("compact,kind,seed", [(True, 16, 31), (False, 32, 2)])
def test_large_compact_structures_sharded_with_logical_ranks(fc, compact, kind, seed):
    """the sharded prune of 260-atom structures -- globules (the split-half screen's 32-column items) and extended ones (the
    fp32 matrix-pipe screen's) -- dealt to three logical ranks in snake order: every rank takes the narrow-tile screen,
    the replayed ladder gives the single-GPU mask, every similar pair is found by exactly one rank"""
    from firecode_amd import _lib

    n, world = 4200, 3
    X, atoms, asg = syn.synthetic_ensemble(n, 260, seed=seed, cluster_size=5, compact=compact)
    counts = []
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            ref, stats0 = ens.prune(0.5, 1.0)
            assert _lib.screen_last_kind() == kind
            for r in list(range(1, world)) + [0]:
                _lib.call("fc_debug_comm_loopback", r, world)
                mask, stats = ens.prune_sharded(0.5, 1.0)
                assert _lib.screen_last_kind() == kind
                counts.append(int(stats[2]))
    finally:
        _lib.call("fc_debug_comm_loopback", -1, 0)
    assert np.array_equal(mask, ref) and mask.sum() == len(np.unique(asg))
    assert int(np.sum(counts)) == stats0[2]


def test_cfg4_full_size_eight_logical_ranks(fc):
    """BASELINE configs[3] whole: 100 000 conformers x 80 atoms, the 8 ranks of the sharded prune
    played on one GPU (row blocks in snake order, one message per rank, replayed ladder):
    one survivor per cluster, messages within capacity and balanced"""
    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    n, world = 100_000, 8
    X, atoms, asg = syn.synthetic_ensemble(n, 80, seed=6)
    n_clusters = len(np.unique(asg))
    cap = fdist.exchange_cap(n, world)
    counts = []
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            del X
            for r in list(range(1, world)) + [0]:  # the ranks one after the other (fc_debug_comm_loopback), C path
                _lib.call("fc_debug_comm_loopback", r, world)
                mask, stats = ens.prune_sharded(0.5, 1.0)
                counts.append(int(stats[2]))
    finally:
        _lib.call("fc_debug_comm_loopback", -1, 0)
    counts = np.array(counts, dtype=np.int64)
    assert mask.sum() == n_clusters
    first = np.zeros(n_clusters, dtype=np.int64)  # the greedy ladder keeps the LAST member of a cluster (i removed when a later j matches)
    np.maximum.at(first, asg, np.arange(n))
    assert np.array_equal(np.flatnonzero(mask), np.sort(first))
    assert counts.max() <= cap and counts.max() - counts.min() < 0.1 * counts.mean()
    assert int(counts.sum()) == int(sum(c * (c - 1) // 2 for c in np.bincount(asg)))


@pytest.mark.parametrize("q", [5, 12])
def test_first_match_at_two_phase_size_without_structure(fc, q):
    """fc_tfd_first_match at a size where the two-phase forms are the default (N >= 65 536) on fingerprints WITHOUT the
    order of a systematic scan (random cluster centres, a third of the rows without any partner): the walk's window
    boxes exclude little, rows stay open for tens of thousands of columns.  16-bit path == fp32 two-phase kernels ==
    the one-phase kernel (Q = 12: the sixteen bits hold a lower bound only, every candidate takes the fp64 sum)."""
    from firecode_amd import _lib as L

    rng = np.random.default_rng(300 + q)
    n = 70000
    centres = rng.uniform(-180, 180, size=(9000, q))
    tf = centres[rng.integers(0, len(centres), n)] + rng.normal(scale=1.2, size=(n, q))
    tf[rng.integers(0, n, n // 3)] = rng.uniform(-180, 180, size=(n // 3, q))
    tf = np.ascontiguousarray((tf + 180) % 360 - 180)
    out = {}
    for name, env in (("u16", {}), ("f32", {"FC_TFD_U16": "0"}), ("one", {"FC_TFD_LOOKAHEAD": "0"})):
        for k in ("FC_TFD_U16", "FC_TFD_LOOKAHEAD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        out[name] = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(tf), n, q, 10.0, L.pi(out[name]))
    for k in ("FC_TFD_U16", "FC_TFD_LOOKAHEAD"):
        os.environ.pop(k, None)
    assert np.array_equal(out["u16"], out["one"]) and np.array_equal(out["f32"], out["one"])
    gap = (out["one"] - np.arange(n))[out["one"] >= 0]
    assert (out["one"] < 0).sum() > n // 4 and (gap > 20000).sum() > 100
    # thresholds at which everything / nothing is similar, and one between two grid values of the sixteen bits
    for thr in (5000.0, 1e-3, 10.0 + 1.0 / 364.0):
        fm = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(tf), n, q, thr, L.pi(fm))
        os.environ["FC_TFD_LOOKAHEAD"] = "0"
        one = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(tf), n, q, thr, L.pi(one))
        os.environ.pop("FC_TFD_LOOKAHEAD", None)
        assert np.array_equal(fm, one), thr
        if thr == 5000.0:
            assert np.array_equal(fm[:-1], np.arange(1, n)) and fm[-1] == -1
        if thr == 1e-3:
            assert (fm < 0).all()
