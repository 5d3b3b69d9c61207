"""GPU parity tests: every HIP path against the oracle on the same seeded
inputs, through the C ABI (firecode_amd -> ctypes -> libfc_hip.so).

Bars (BASELINE.json north_star): integer / boolean results bit-exact;
RMSD, max deviation and coordinates within 1e-10.
"""

import os

import numpy as np
import pytest

from firecode_amd import synthetic as syn
from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _rot(rng):
    return syn.random_rotation(rng)


# ---------------------------------------------------------------- a4 / a9
def test_rmsd_and_max_pairs(fc):
    X, atoms, _ = syn.synthetic_ensemble(120, 50, seed=3)
    iu, ju = np.triu_indices(len(X), 1)
    r, m = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True)
    r0, m0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(r - r0).max() < TOL
    # max deviation: 1e-10, widened only by the pair's own conditioning (oracle.rotation_error_bound_batch:
    # (2A + 8) eps (Gp + Gq)/2 / (f2 + f3) r_max -- how far two correct float64 evaluations may differ)
    bound = o.rotation_error_bound_batch(X[iu], X[ju], center=True)
    assert np.all(np.abs(m - m0) <= TOL + bound)
    assert (bound < TOL).mean() > 0.99  # ... which is below the tolerance itself for nearly every pair


def test_rmsd_and_max_single_no_center(fc):
    rng = np.random.default_rng(5)
    p = rng.normal(size=(33, 3))
    q = p @ _rot(rng).T + 0.01 * rng.normal(size=p.shape)
    for center in (False, True):
        r, m = fc.rmsd.rmsd_and_max(p, q, center=center)
        r0, m0 = o.rmsd_and_max(p, q, center=center)
        assert abs(r - r0) < TOL and abs(m - m0) < TOL
    # identical structures, pure rotation about the origin
    r, m = fc.rmsd.rmsd_and_max(p, p @ _rot(rng).T)
    assert r < 1e-12 and m < 1e-12


def test_rmsd_matrix(fc):
    X, atoms, _ = syn.synthetic_ensemble(200, 30, seed=4)
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D = ens.rmsd_matrix()
    S0, R0, D0 = o.rmsd_similarity_matrix(X, atoms, 0.5)
    assert np.abs(R - R0).max() < TOL
    assert np.allclose(R, R.T) and np.all(np.diag(R) == 0)
    assert np.abs(D - D0)[R0 < 1.0].max() < TOL


def test_heavy_atom_mask(fc):
    X, _, _ = syn.synthetic_ensemble(40, 24, seed=6)
    atoms = np.array(["C", "H", "H", "O"] * 6)
    hv = atoms != "H"
    iu, ju = np.triu_indices(len(X), 1)
    r, m = fc.rmsd.rmsd_and_max_batch(X, iu, ju, center=True, atom_mask=hv)
    r0, m0 = o.rmsd_and_max_batch(X[iu][:, hv], X[ju][:, hv], center=True)
    assert np.abs(r - r0).max() < TOL


def test_alignment_matrix_and_align_vec_pair(fc, golden):
    rng = np.random.default_rng(8)
    P = rng.normal(size=(50, 20, 3))
    Q = np.array([p @ _rot(rng).T for p in P]) + 0.05 * rng.normal(size=P.shape)
    M = fc.rmsd.get_alignment_matrices(P, Q)
    M0 = np.array([o.get_alignment_matrix(p, q) for p, q in zip(P, Q)])
    assert np.abs(M - M0).max() < TOL
    assert np.abs(np.linalg.det(M) - 1).max() < 1e-12
    # in-tree twin, against the reference's own outputs (rank-2 cases)
    out = fc.algebra.align_vec_pair_batch(golden["avp_ref"], golden["avp_tgt"])
    ok = np.ones(len(out), dtype=bool)
    ok[8:12] = False  # rank-1 covariance: rotation not unique
    assert np.abs(out - golden["avp_out"])[ok].max() < TOL
    one = fc.algebra.align_vec_pair(golden["avp_ref"][20], golden["avp_tgt"][20])
    assert np.abs(one - golden["avp_out"][20]).max() < TOL


def test_align_structures(fc):
    X, atoms, _ = syn.synthetic_ensemble(30, 25, seed=9)
    out = fc.utils.align_structures(X)
    assert np.abs(out - o.align_structures(X)).max() < TOL
    idx = np.array([0, 3, 5, 8, 13, 21])
    out = fc.utils.align_structures(X, idx)
    assert np.abs(out - o.align_structures(X, idx)).max() < TOL


# ---------------------------------------------------------------- a5 / a6
@pytest.mark.parametrize("n,a,seed", [(64, 12, 1), (257, 30, 2), (600, 50, 3), (1000, 17, 4)])
def test_prune_by_rmsd_mask_bit_exact(fc, n, a, seed):
    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=seed)
    S0, R0, D0 = o.rmsd_similarity_matrix(X, atoms, 0.5)
    assert np.abs(R0[R0 > 0] - 0.5).min() > 1e-6  # generator keeps pairs off the threshold
    ref_mask = o.greedy_prune_from_matrix(S0)
    pruned, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert mask.dtype == np.bool_ and mask.shape == (n,)
    assert np.array_equal(mask, ref_mask)
    assert np.array_equal(pruned, X[mask])
    assert mask.sum() == len(np.unique(asg))


def test_prune_many_equals_one_by_one_and_oracle(fc):
    """a queue of ensembles of different sizes (one structure, empty, below and above one row
    block, one where everything is similar) pruned in flight together: each mask is the mask
    of its own prune_by_rmsd call and of the oracle"""
    rng = np.random.default_rng(5)
    queue = []
    for n, a, seed in [(300, 20, 21), (1, 9, 22), (64, 12, 23), (700, 33, 24), (129, 50, 25), (257, 8, 26)]:
        X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
        queue.append((X, atoms))
    queue.insert(2, (np.zeros((0, 7, 3)), np.array(["C"] * 7)))
    same = np.repeat(rng.normal(size=(1, 15, 3)), 90, axis=0) + rng.normal(scale=1e-3, size=(90, 15, 3))
    queue.append((same, np.array(["C"] * 15)))
    many = fc.pruner.prune_many_by_rmsd(queue, 0.5)
    assert len(many) == len(queue)
    for (X, atoms), (pruned, mask) in zip(queue, many):
        _, single = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
        assert np.array_equal(mask, single)
        assert np.array_equal(pruned, X[mask])
        if 1 < len(X) <= 300:
            S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
            assert np.array_equal(mask, o.greedy_prune_from_matrix(S0))
    assert many[-1][1].sum() == 1
    # the same queue again: workspaces of the first call are gone, results are not
    again = fc.pruner.prune_many_by_rmsd(queue[:3], 0.5)
    for (_, m0), (_, m1) in zip(many[:3], again):
        assert np.array_equal(m0, m1)


def test_prune_many_rejects_a_repeated_ensemble(fc):
    X, atoms, _ = syn.synthetic_ensemble(80, 10, seed=3)
    from firecode_amd import _lib as L

    with fc.DeviceEnsemble(X, center=True) as ens:
        with pytest.raises(fc.FirecodeHipInputError):
            L.prune_many([ens, ens], 0.5, 1.0)
        masks, alive = L.prune_many([ens], 0.5, 1.0)
        assert alive[0] == masks[0].sum()


def test_prune_by_rmsd_literal_oracle_small(fc):
    """against the literal (sequential, cached) restatement, not the matrix form"""
    X, atoms, _ = syn.synthetic_ensemble(150, 20, seed=12)
    _, ref = o.prune_by_rmsd(X, atoms, 0.5)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask, ref)


def test_simbits_match_oracle_matrix(fc):
    X, atoms, _ = syn.synthetic_ensemble(300, 40, seed=13)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, grey = ens.simbits(0.5, 1.0)
    from firecode_amd._lib import unpack_bits

    S = unpack_bits(bits, len(X))
    assert np.array_equal(S, np.triu(S0, 1))
    assert grey == 0


@pytest.mark.parametrize("n,a,seed,thr", [(500, 50, 31, 0.5), (400, 13, 32, 0.35), (300, 90, 33, 0.25), (257, 30, 34, 1.2)])
def test_fp32_screen_and_fp64_screen_give_the_same_similarity_bits(fc, monkeypatch, n, a, seed, thr):
    """the single-precision screen (fp32 MFMA + bounded polynomial) may only ADD candidates for
    the exact refine: similarity bits, grey count and the pruned mask are those of the fp64 screen
    and of the oracle matrix.  Thresholds are also put exactly on, one ulp below and one ulp above
    the RMSD of real pairs (the band the fp32 test cannot decide), with the band check switched
    off (FC_SCREEN_F32=2) so that the fp32 kernel runs whatever the geometry"""
    from firecode_amd._lib import unpack_bits

    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
    S0, R0, D0 = o.rmsd_similarity_matrix(X, atoms, thr)
    near = np.sort(R0[np.triu_indices(n, 1)])
    k = int(np.searchsorted(near, thr))
    picks = [thr] + [float(v) for v in near[max(k - 2, 0): k + 2]]
    picks += [float(np.nextafter(picks[1], 0.0)), float(np.nextafter(picks[1], 9.0))]
    for t in picks:
        ref = np.triu((R0 < t) & (D0 < 2 * t), 1)
        got = {}
        from firecode_amd import _lib

        # "2" / "3": the single-precision screen of the launcher's choice (split-half f16 pipe at these sizes),
        # alone / with the verdict; "k16" / "k32": each of the two single-precision kernels by name
        for mode in ("0", "2", "3", "k16", "k32"):
            if mode.startswith("k"):
                monkeypatch.delenv("FC_SCREEN_F32", raising=False)
                _lib.screen_select(int(mode[1:]))
            else:
                monkeypatch.setenv("FC_SCREEN_F32", mode)
            try:
                with fc.DeviceEnsemble(X, center=True) as ens:
                    bits, grey = ens.simbits(t, 2 * t)
                    kind_bits = _lib.screen_last_kind()
                    mask, stats = ens.prune(t, 2 * t)
                    kind_lean = _lib.screen_last_kind()
            finally:
                _lib.screen_select(0)
            if mode.startswith("k"):
                assert kind_bits == kind_lean == int(mode[1:])
            got[mode] = (unpack_bits(bits, n), grey, mask, int(stats[2]))
        for mode in ("2", "3", "k16", "k32"):
            assert np.array_equal(got["0"][0], got[mode][0]) and got["0"][1] == got[mode][1]
            assert np.array_equal(got["0"][2], got[mode][2]) and got["0"][3] == got[mode][3]
        if t == thr:  # off-threshold by construction: the oracle's own arithmetic agrees as well
            assert np.abs(near - thr).min() > 1e-7
            assert np.array_equal(got["2"][0], ref)
            assert np.array_equal(got["2"][2], o.greedy_prune_from_matrix(ref | ref.T))
    monkeypatch.delenv("FC_SCREEN_F32", raising=False)


def test_fp32_screen_far_from_the_origin(fc, monkeypatch):
    """uncentred structures 40 A from the origin (G large against A * thr^2): the fp32 test can
    rule out next to nothing, every pair becomes a candidate, the result must not change; by
    default the launcher sees the wide band and takes the fp64 screen"""
    rng = np.random.default_rng(41)
    bases = rng.normal(scale=2.5, size=(40, 1, 20, 3))  # 40 shapes x 8 noisy copies, no rotation between copies
    X = (bases + rng.normal(scale=0.05, size=(40, 8, 20, 3))).reshape(320, 20, 3) + np.array([40.0, -25.0, 10.0])
    from firecode_amd._lib import unpack_bits

    out = {}
    for mode in ("0", "2", "3", None):  # fp64, fp32, fp32 + verdict (-> fp64 redo on the device), the launcher's own choice
        if mode is None:
            monkeypatch.delenv("FC_SCREEN_F32")
        else:
            monkeypatch.setenv("FC_SCREEN_F32", mode)
        with fc.DeviceEnsemble(X, center=False) as ens:
            bits, grey = ens.simbits(0.5, 1.0)
            out[mode] = (unpack_bits(bits, len(X)), grey)
        from firecode_amd import _lib

        # which screen was launched (first): forced by the knob; the launcher's own band estimate finds the split-half
        # band (2.0 A^2) too wide and either takes the fp64 screen at once or, since the fp32 kernel's bound is tighter
        # (0.7 A^2 here, below four squared thresholds), that kernel with the verdict behind it
        assert _lib.screen_last_kind() in {"0": (64,), "2": (16, 32), "3": (16, 32), None: (64, 32)}[mode]
    assert np.array_equal(out["0"][0], out["2"][0]) and np.array_equal(out["0"][0], out[None][0])
    assert np.array_equal(out["0"][0], out["3"][0])
    assert out["0"][1] == out["2"][1] == out["3"][1] == out[None][1]
    assert out["0"][0].any()
    # the same verdict inside a pipeline of prunes (three lanes): no fp64 launch in place -- the verdict empties the
    # queues, the pair ladder declines, the call redoes the prune synchronously; same mask as the fp64 screen's
    masks = {}
    for mode in ("0", "3"):
        monkeypatch.setenv("FC_SCREEN_F32", mode)
        with fc.DeviceEnsemble(X, center=False) as ens:
            _, _, masks[mode], st = ens.bench_prune(0.5, 1.0, reps=7, want_mask=True)
            _, _, again, _ = ens.bench_prune(0.5, 1.0, reps=4, want_mask=True)
            assert np.array_equal(masks[mode], again)
    assert np.array_equal(masks["0"], masks["3"]) and 0 < masks["0"].sum() < len(X)
    S = out["0"][0]
    assert np.array_equal(masks["0"], o.greedy_prune_from_matrix(S | S.T))
    monkeypatch.delenv("FC_SCREEN_F32")
    # a compact, centred ensemble of the bench's kind: the launcher takes the single-precision screen
    Xc, _, _ = syn.synthetic_ensemble(300, 50, seed=5)
    with fc.DeviceEnsemble(Xc, center=True) as ens:
        ens.simbits(0.5, 1.0)
    assert _lib.screen_last_kind() in (16, 32)


@pytest.mark.parametrize("n,a,seed,thr", [(2200, 30, 5, 0.75), (1500, 23, 9, 0.9), (1500, 100, 12, 0.9), (1500, 104, 11, 0.9)])
def test_long_candidate_queue_refine_vs_oracle(fc, n, a, seed, thr):
    """More than 2^17 candidates per prune: the queue is ordered by (128-row block, 64-column tile) bucket
    (k_bucket_count / _scan / _scatter) and refined bucket by bucket with the column tile in LDS (k_refine_buckets;
    odd atom count: padded row; 100 atoms: the largest column tile that fits LDS beside the kernel's own arrays; 104: the
    tile alone would fit, tile + arrays do not, and the queue is walked as it lies).  Every other test's queue is short
    enough for the 8-lanes-per-pair kernel.
    Similarity bits (the refine's atomicAnd path), similar-pair count, grey count and masks with and without an
    energy window against the oracle's all-pairs matrix."""
    from firecode_amd._lib import unpack_bits

    X = syn.continuous_ensemble(n, a, seed=seed)
    atoms = np.array(["C"] * a)
    S0, R0, D0 = o.rmsd_similarity_matrix(X, atoms, thr)
    iu = np.triu_indices(n, 1)
    assert np.abs(R0[iu] - thr).min() > 1e-9 and np.abs(D0[iu] - 2 * thr)[R0[iu] < thr].min() > 1e-9
    en = np.random.default_rng(seed).uniform(0, 3, size=n)
    with fc.DeviceEnsemble(X, center=True) as ens:
        # the first call over these coordinates takes the straight queue walk (k_refine_pairs); from then on the library
        # knows the queue is long and orders it by bucket first -- both forms, with and without the bit matrix
        bits_w, grey_w = ens.simbits(thr, 2 * thr)
        mask_w, stats_w = ens.prune(thr, 2 * thr)
        bits, grey = ens.simbits(thr, 2 * thr)
        mask, stats = ens.prune(thr, 2 * thr)
        _, _, mask_p, stats_p = ens.bench_prune(thr, 2 * thr, reps=5, want_mask=True)  # ... and inside the pipelined prunes
    assert int(stats[1]) > (1 << 17) and int(stats[2]) == int(np.triu(S0, 1).sum()) > (1 << 17)
    assert np.array_equal(unpack_bits(bits, n), np.triu(S0, 1)) and grey == 0
    assert np.array_equal(bits, bits_w) and grey_w == 0 and np.array_equal(mask, mask_w) and np.array_equal(stats[1:4], stats_w[1:4])
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S0)) and np.array_equal(mask_p, mask) and int(stats_p[2]) == int(stats[2])
    _, mask_e = fc.pruner.prune_by_rmsd(X, atoms, thr, energies=en, max_dE=1.0)  # (sorts by energy, as the reference does)
    assert np.array_equal(mask_e, o.greedy_prune_from_matrix(S0, energies=en, max_dE=1.0))
    assert 0 < mask.sum() < mask_e.sum() < n


def test_wide_band_takes_the_speculative_screen(fc):
    """An 80-atom skeleton that happens to be stretched (radius of gyration 10.7 A): the band of mean square deviations
    the bounded single-precision test cannot decide is 0.265 A^2 -- wider than the squared threshold, and EMPTY in a
    clustered ensemble.  Until round 4 such ensembles went to the fp64 screen at once (5 x the time); now the split-half
    screen runs and the device-side verdict stays silent.  Mask against the oracle; then the same coordinates with a
    band that IS populated (a continuous spread below 2.2 thresholds): whatever the verdict says, the mask is the oracle's."""
    from firecode_amd import _lib

    X, atoms, _ = syn.synthetic_ensemble(600, 80, seed=4)
    G = ((X - X.mean(axis=1, keepdims=True)) ** 2).sum(axis=(1, 2)).max()
    assert 0.25 < 2 * 1.1643e-3 * G / 80 < 1.0   # the band of kabsch_h2_bounds(3): between 1 and 4 squared thresholds
    S0, R0, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    with fc.DeviceEnsemble(X, center=True) as ens:
        mask, stats = ens.prune(0.5, 1.0)
        assert _lib.screen_last_kind() == 16
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S0)) and int(stats[1]) == int(np.triu(S0, 1).sum())
    rng = np.random.default_rng(8)
    Y = X[:1] + rng.normal(size=(600, 1, 1)) * rng.normal(scale=0.12, size=(1, 80, 3)) + rng.normal(scale=0.05, size=(600, 80, 3))
    S1, R1, D1 = o.rmsd_similarity_matrix(Y, atoms, 0.5)
    iu = np.triu_indices(600, 1)
    assert np.abs(R1[iu] - 0.5).min() > 1e-9 and ((R1[iu] > 0.5) & (R1[iu] < 0.7)).sum() > 1000  # a populated band
    _, mask1 = fc.pruner.prune_by_rmsd(Y, atoms, 0.5)
    assert np.array_equal(mask1, o.greedy_prune_from_matrix(S1))


def test_random_prunes_vs_oracle(fc):
    """A seeded sweep the directed cases do not cover one by one: 2 - 700 conformers, 1 - 140 atoms (hydrogens mixed in:
    the heavy-atom mask), clustered / continuous / unrelated / duplicated / planar / mirrored structures with random
    rigid motions, random thresholds, with and without an energy window -- the mask of ``prune_by_rmsd`` against the
    oracle's greedy ladder over its own all-pairs matrix.  A case with a pair within 1e-9 of a threshold is skipped
    (the decision is then a matter of the last bits of two different solvers); at least 25 of the 36 must count."""
    rng = np.random.default_rng(20261004)
    counted = 0
    for case in range(36):
        n = int(rng.choice([2, 3, 5, 17, 63, 64, 65, 100, 129, 200, 257, 400, 511, 700]))
        a = int(rng.choice([1, 2, 3, 4, 7, 16, 23, 33, 50, 64, 65, 97, 128, 140]))
        kind = str(rng.choice(["clusters", "continuous", "random", "duplicates", "planar", "mirror"]))
        seed = int(rng.integers(1 << 30))
        if kind == "clusters" and a >= 3:
            X = syn.synthetic_ensemble(max(n, 5), a, seed=seed)[0][:n]
        elif kind == "continuous" and a >= 3:
            X = syn.continuous_ensemble(n, a, seed=seed)
        else:
            X = rng.normal(scale=1.5, size=(n, a, 3))
            if kind == "duplicates":
                X[n // 2:] = X[: n - n // 2] + rng.normal(scale=0.02, size=(n - n // 2, a, 3))
            elif kind == "planar":
                X[:, :, 2] = 0.0
                X[1::2] = X[0::2][: len(X[1::2])] + rng.normal(scale=0.05, size=(len(X[1::2]), a, 3)) * np.array([1.0, 1.0, 0.0])
            elif kind == "mirror":
                X[1::2] = X[0::2][: len(X[1::2])] * np.array([1.0, 1.0, -1.0])
        X = np.einsum("nij,naj->nai", np.array([_rot(rng) for _ in range(n)]), X) + rng.normal(scale=3.0, size=(n, 1, 3))
        atoms = np.where(rng.random(a) < 0.25, "H", "C")
        if not (atoms != "H").any():
            atoms[0] = "C"
        thr = float(rng.choice([0.05, 0.25, 0.5, 1.0, 2.0]))
        S0, R0, D0 = o.rmsd_similarity_matrix(X, atoms, thr)
        iu = np.triu_indices(n, 1)
        if np.abs(R0[iu] - thr).min() < 1e-9 or (np.abs(D0[iu] - 2 * thr)[R0[iu] < thr] < 1e-9).any():
            continue
        en = rng.uniform(0, 3, size=n) if rng.random() < 0.4 else None
        ref = o.greedy_prune_from_matrix(S0, energies=en, max_dE=1.0) if en is not None else o.greedy_prune_from_matrix(S0)
        kw = {"energies": en, "max_dE": 1.0} if en is not None else {}
        pruned, mask = fc.pruner.prune_by_rmsd(X, atoms, thr, **kw)
        assert np.array_equal(mask, ref), (case, n, a, kind, thr, en is not None)
        assert pruned.shape == (int(ref.sum()), a, 3) and np.array_equal(pruned, X[ref])
        counted += 1
    assert counted >= 25


def test_refine_word_queue_fallback(fc, monkeypatch):
    """pair queue too small -> the refine kernel must fall back to the word
    queue (sparse words: wave per pair; dense words: lane per pair)"""
    monkeypatch.setenv("FC_PAIRQ_CAP", "4")
    X, atoms, _ = syn.synthetic_ensemble(300, 20, seed=16)  # sparse words
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S0))
    X, atoms, _ = syn.synthetic_ensemble(256, 15, seed=17, cluster_size=64)  # dense words
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, _ = ens.simbits(0.5, 1.0)
    from firecode_amd._lib import unpack_bits

    assert np.array_equal(unpack_bits(bits, len(X)), np.triu(S0, 1))
    assert np.triu(S0, 1).sum() > 4000


def test_prune_with_energies(fc):
    X, atoms, _ = syn.synthetic_ensemble(300, 20, seed=14)
    rng = np.random.default_rng(14)
    en = rng.uniform(0, 3, size=len(X))
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0, energies=en, max_dE=1.0)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0)
    assert np.array_equal(mask, ref)


def test_prune_edge_cases(fc):
    atoms = np.array(["C"] * 7)
    # empty ensemble
    p, m = fc.pruner.prune_by_rmsd(np.zeros((0, 7, 3)), atoms, 0.5)
    assert p.shape == (0, 7, 3) and m.shape == (0,)
    # single structure
    X = np.random.default_rng(0).normal(size=(1, 7, 3))
    p, m = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert m.tolist() == [True]
    # all identical up to rigid motion: the last one survives
    rng = np.random.default_rng(1)
    base = rng.normal(scale=2.0, size=(7, 3))
    X = np.array([base @ _rot(rng).T + rng.normal(size=3) for _ in range(70)])
    _, m = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    _, ref = o.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(m, ref) and m.sum() == 1 and m[-1]
    # all distinct
    X = rng.normal(scale=3.0, size=(65, 7, 3))
    _, m = fc.pruner.prune_by_rmsd(X, atoms, 0.05)
    assert m.all()


def test_greedy_from_bits(fc):
    rng = np.random.default_rng(3)
    n = 700
    S = np.triu(rng.random((n, n)) < 0.002, 1)
    W = (n + 63) // 64
    padded = np.zeros((n, W * 64), dtype=np.uint8)
    padded[:, :n] = S
    bits = np.packbits(padded, axis=1, bitorder="little").view(np.uint64)
    mask = fc.pruner.greedy_prune_from_bits(bits, n)
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S | S.T))


def test_moments_and_moi_prune(fc):
    X, atoms, _ = syn.synthetic_ensemble(260, 22, seed=15)
    atoms = np.array((["C", "H", "N", "O", "H"] * 5)[:22])
    masses = np.array([o.MASSES_TABLE[a] for a in atoms])
    mom = fc.algebra.get_inertia_moments_batch(X, masses)
    mom0 = np.array([o.get_inertia_moments(x, masses) for x in X])
    assert np.abs(mom - mom0).max() < 1e-13 * mom0.max()  # amu A^2 of size 1e3..1e5: relative, ~500 eps (symmetric eigenvalues are perfectly conditioned)
    _, ref = o.prune_by_moment_of_inertia(X, atoms)
    _, mask = fc.pruner.prune_by_moment_of_inertia(X, atoms)
    assert np.array_equal(mask, ref)
    assert 0 < mask.sum() < len(mask)


# ---------------------------------------------------------------- a11 / a12 / a13
def test_count_clashes_golden(fc, golden):
    out = fc.algebra.count_clashes_batch(golden["cc_in"])
    assert np.array_equal(out, golden["cc_out"])
    assert fc.algebra.count_clashes(golden["cc_in"][0]) == golden["cc_out"][0]


def test_compenetration_check_golden(fc, golden):
    cp = golden["cp_in"]
    assert np.array_equal(fc.utils.compenetration_check_batch(cp), golden["cp_none"])
    assert np.array_equal(fc.utils.compenetration_check_batch(cp, max_clashes=2), golden["cp_none_mc2"])
    for thr in (1.0, 1.5):
        for mc in (0, 3):
            bi = fc.utils.compenetration_check_batch(cp, ids=[20, 16], thresh=thr, max_clashes=mc)
            tri = fc.utils.compenetration_check_batch(cp, ids=[12, 14, 10], thresh=thr, max_clashes=mc)
            assert np.array_equal(bi, golden[f"cp_bi_{thr}_{mc}"])
            assert np.array_equal(tri, golden[f"cp_tri_{thr}_{mc}"])
    assert fc.utils.compenetration_check(cp[3], ids=[20, 16], thresh=1.5) == bool(golden["cp_bi_1.5_0"][3])


def test_clash_threshold_ties(fc):
    """distances exactly at / next to the threshold decide like cdist does"""
    thr = 1.5
    xs = [thr, np.nextafter(thr, 0), np.nextafter(thr, 9), 0.3 * 5, 1.4999999999999998]
    coords = np.array([[[0, 0, 0], [x, 0, 0]] for x in xs] +
                      [[[0.1, 0.2, 0.3], [0.1 + x / np.sqrt(3), 0.2 + x / np.sqrt(3), 0.3 + x / np.sqrt(3)]] for x in xs])
    for ids, kw in (([1, 1], {}),):
        got = fc.utils.compenetration_check_batch(coords, ids=ids, thresh=thr)
        ref = [o.compenetration_check(c, ids=ids, thresh=thr) for c in coords]
        assert np.array_equal(got, ref)
    tri = np.concatenate([coords, coords[:, :1] + 50.0], axis=1)
    got = fc.utils.compenetration_check_batch(tri, ids=[1, 1, 1], thresh=thr)
    ref = [o.compenetration_check(c, ids=[1, 1, 1], thresh=thr) for c in tri]
    assert np.array_equal(got, ref)


def test_get_embed_and_rototranslate(fc, golden):
    class Mol:
        pass

    m1, m2 = Mol(), Mol()
    m1.coords, m2.coords = golden["ge_c1"], golden["ge_c2"]
    for R, t, ids, exp in zip(golden["ge_R"], golden["ge_t"], golden["ge_ids"], golden["ge_out"]):
        m1.rotation, m1.position = R[0], t[0]
        m2.rotation, m2.position = R[1], t[1]
        out = fc.embeds.get_embed([m1, m2], ids)
        assert np.abs(out - exp).max() < TOL


def test_embed_poses_clash(fc):
    rng = np.random.default_rng(21)
    m1 = rng.normal(scale=2.0, size=(6, 40, 3))
    m2 = rng.normal(scale=2.0, size=(5, 37, 3))
    P = 500
    c1, c2 = rng.integers(0, 6, P), rng.integers(0, 5, P)
    R1 = np.array([_rot(rng) for _ in range(P)])
    R2 = np.array([_rot(rng) for _ in range(P)])
    t1 = rng.normal(scale=1.0, size=(P, 3))
    t2 = t1 + rng.normal(scale=4.0, size=(P, 3))
    ok, counts, poses = fc.embeds.embed_poses_clash(m1, m2, c1, c2, R1, t1, R2, t2, thresh=1.5,
                                                    max_clashes=0, return_poses=True)
    ref_pose = np.array([o.get_embed([m1[a], m2[b]], [ra, rb], [ta, tb])
                         for a, b, ra, rb, ta, tb in zip(c1, c2, R1, R2, t1, t2)])
    assert np.abs(poses - ref_pose).max() < TOL
    ref_ok = np.array([o.compenetration_check(p, ids=[40, 37], thresh=1.5) for p in ref_pose])
    assert np.array_equal(ok, ref_ok)
    assert 0 < ok.sum() < P
    ok2, counts2 = fc.embeds.embed_poses_clash(m1, m2, c1, c2, R1, t1, R2, t2, thresh=1.5, max_clashes=4)
    ref_ok2 = np.array([o.compenetration_check(p, ids=[40, 37], thresh=1.5, max_clashes=4) for p in ref_pose])
    assert np.array_equal(ok2, ref_ok2) and np.array_equal(counts, counts2)


# ---------------------------------------------------------------- a17 - a20
def _chain_case(n_atoms, n_tors, seed):
    """zig-zag chain with side atoms; torsions about backbone bonds, mask = the
    atoms after the bond (what _get_rotation_mask returns for a chain)."""
    rng = np.random.default_rng(seed)
    skel = syn.synthetic_skeleton(n_atoms, rng)
    centres = np.linspace(2, n_atoms - 4, n_tors).astype(int)
    torsions, masks = [], []
    for c in centres:
        torsions.append((c - 1, c, c + 1, c + 2))
        m = np.zeros(n_atoms, dtype=bool)
        m[c + 2:] = True
        masks.append(m)
    return skel, np.array(torsions), np.array(masks)


def test_pinned_host_arrays_are_taken_like_any_other(fc):
    """fc_host_alloc_pinned behind ``firecode_amd.pinned_empty``: an ensemble built into page-locked memory is uploaded by
    direct DMA (no staging copy) and pruned to the same mask as from an ordinary array; views keep the block alive, the
    last one frees it; a zero-size request is an ordinary empty array."""
    import gc

    X, atoms, asg = syn.synthetic_ensemble(3000, 50, seed=12)
    _, ref = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    P = fc.pinned_empty(X.shape)
    assert P.shape == X.shape and P.dtype == np.float64 and P.flags.c_contiguous and P.flags.writeable
    P[...] = X
    view = P[100:200]
    _, mask = fc.pruner.prune_by_rmsd(P, atoms, 0.5)
    assert np.array_equal(mask, ref)
    del P
    gc.collect()
    assert np.array_equal(view, X[100:200])  # (the view still owns the block)
    _, m2 = fc.pruner.prune_by_rmsd(np.ascontiguousarray(view), atoms, 0.5)
    assert m2.shape == (100,)
    del view
    gc.collect()
    assert fc.pinned_empty((0, 5, 3)).shape == (0, 5, 3)
    for _ in range(20):  # (blocks come and go: nothing accumulates, nothing is freed twice)
        Q = fc.pinned_empty((64, 50, 3))
        Q[...] = X[:64]
        del Q
    gc.collect()
    _, m3 = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(m3, ref)


def test_torsion_scan_vs_oracle(fc):
    base, tors, masks = _chain_case(30, 4, seed=31)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4)[::5]
    out, rot = fc.torsion_module.torsion_scan(base, tors, masks, angles, thresh=1.5)
    out0, rot0 = o.torsion_scan(base, tors, masks, angles, thresh=1.5)
    assert np.array_equal(rot, rot0)
    assert np.abs(out - out0).max() < TOL
    assert (rot0 < (angles != 0).sum(axis=1)).any()  # some back-off / failed rotations happened


@pytest.mark.parametrize("thresh,backoff", [(1.5, 5), (2.0, 5), (2.0, 7), (1.5, 2)])
def test_scan_back_off_closed_form_equals_the_walked_loop(fc, monkeypatch, thresh, backoff):
    """The scan tree finds the step at which a back-off loop (torsion_module.py:831-842) ends from the closed form of
    the pair distances under rotation and only then turns the atoms (fc_torsion.hip: torsion_step).  Same coordinates
    bit for bit, same counts as the loop walked step by step (FC_SCAN_CLOSED_FORM=0) and as the one-wavefront-per-row
    kernel (FC_SCAN_TREE=0), on scans where most nodes clash and a third to two thirds of the loops run out, with a
    back-off that does not divide the angles (7) and with one that needs more steps than the kernel's table holds
    (2 degrees: 150 steps, the loop); a sample against the oracle."""
    base, tors, masks = _chain_case(40, 5, seed=72)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 5)  # 7 776 angle-sets: the tree takes them
    runs = {}
    for name, env in (("closed", {}), ("loop", {"FC_SCAN_CLOSED_FORM": "0"}), ("rows", {"FC_SCAN_TREE": "0"})):
        for k in ("FC_SCAN_CLOSED_FORM", "FC_SCAN_TREE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if name == "rows":
            runs[name] = fc.torsion_module.torsion_scan(base, tors, masks, angles, thresh=thresh, backoff=backoff)
        else:
            tf, rot, out = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, tors, thresh=thresh,
                                                                       backoff=backoff, want_coords=True)
            runs[name] = (out, rot, tf)
    for k in ("FC_SCAN_CLOSED_FORM", "FC_SCAN_TREE"):
        monkeypatch.delenv(k, raising=False)
    assert np.array_equal(runs["closed"][0], runs["loop"][0]) and np.array_equal(runs["closed"][1], runs["loop"][1])
    assert np.array_equal(runs["closed"][2], runs["loop"][2])
    assert np.array_equal(runs["closed"][0], runs["rows"][0]) and np.array_equal(runs["closed"][1], runs["rows"][1])
    pick = np.random.default_rng(5).integers(0, len(angles), 120)
    ref_c, ref_r = o.torsion_scan(base, tors, masks, angles[pick], thresh=thresh, backoff=backoff)
    assert np.array_equal(runs["closed"][1][pick], ref_r)
    assert np.abs(runs["closed"][0][pick] - ref_c).max() < TOL
    nz = (angles != 0).sum(axis=1)
    assert 0.1 < (runs["closed"][1] < nz).mean() < 0.9  # back-off loops that ran out, and loops that ended early


@pytest.mark.parametrize("n_atoms,n_quads", [(90, 5), (140, 5), (40, 70), (64, 33)])
def test_scan_tree_size_classes_of_the_level_kernel(fc, monkeypatch, n_atoms, n_quads):
    """k_ts_level's compile-time classes: a node's state through 3 x 64 registers (<= 64 atoms), 6 x 64 (<= 128) or read in
    place (more); the last level's dihedrals 64 / Q conformers at a time (Q <= 64) or straight away (more); the closed
    form of the back-off loop for <= 256 (rest, moving) pairs per torsion, the walked loop beyond (the early torsions
    of a 90-atom chain have thousands).  Tree == one-wavefront-per-row kernel bit for bit (coordinates, counts,
    fingerprints), a sample against the oracle."""
    base, tors, masks = _chain_case(n_atoms, 5, seed=200 + n_atoms)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 5)
    rng = np.random.default_rng(n_atoms + n_quads)
    quads = tors if n_quads == 5 else np.array([rng.choice(n_atoms, 4, replace=False) for _ in range(n_quads)])
    runs = {}
    for name, env in (("tree", {}), ("rows", {"FC_SCAN_TREE": "0"})):
        monkeypatch.delenv("FC_SCAN_TREE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tf, rot, out = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, quads, thresh=1.6, want_coords=True)
        runs[name] = (out, rot, tf)
    monkeypatch.delenv("FC_SCAN_TREE", raising=False)
    for a, b in zip(runs["tree"], runs["rows"]):
        assert np.array_equal(a, b)
    pick = rng.integers(0, len(angles), 60)
    ref_c, ref_r = o.torsion_scan(base, tors, masks, angles[pick], thresh=1.6)
    assert np.array_equal(runs["tree"][1][pick], ref_r)
    assert np.abs(runs["tree"][0][pick] - ref_c).max() < TOL
    assert np.abs(runs["tree"][2][pick] - o.get_tf_mat(ref_c, quads)).max() < 1e-8  # (arbitrary quadruplets: dihedrals near +-180 amplify)


def test_scan_back_off_threshold_exactly_on_a_distance(fc):
    """A clash threshold EQUAL to a distance the back-off loop meets: the closed form cannot decide such a pair (its
    value is within roundings of the threshold) and hands the node to the walked loop; counts as the oracle's
    (`cdist < thresh`, strict), coordinates within 1e-10."""
    from scipy.spatial.distance import cdist

    base, tors, masks = _chain_case(40, 5, seed=74)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 5)
    hit = 0
    for t, angle, thr0 in [(t, a, h) for h in (2.0, 1.6, 2.4) for t in (4, 3, 2, 1, 0) for a in (120, 60, 180, 240, 300)]:
        if hit == 3:
            break
        # walk this torsion's back-off by hand for the angle-set that turns only it, take a distance of step 2 as threshold
        mask = masks[t].astype(bool)
        temp = o.rotate_dihedral(base, tors[t], angle, mask)
        if o.torsion_comp_check(temp, tors[t], mask, thr0):
            continue  # (no clash to back off from)
        for _ in range(2):
            temp = o.rotate_dihedral(temp, tors[t], -5, mask)
        anti = ~mask
        anti[tors[t][1]] = anti[tors[t][2]] = False
        d = cdist(temp[anti], temp[mask])
        below = np.sort(d[d < thr0])
        if len(below) == 0:
            continue
        thresh = float(below[0])  # the closest pair of step 2 sits exactly on the threshold
        hit += 1
        row = np.zeros(5, dtype=angles.dtype)
        row[t] = angle
        out, rot = fc.torsion_module.torsion_scan(base, tors, masks, angles, thresh=thresh)  # (tree: 7 776 rows)
        pick = np.concatenate([[int(np.flatnonzero((angles == row).all(axis=1))[0])],
                               np.random.default_rng(6).integers(0, len(angles), 40)])
        ref_c, ref_r = o.torsion_scan(base, tors, masks, angles[pick], thresh=thresh)
        assert np.array_equal(rot[pick], ref_r), (t, angle)
        assert np.abs(out[pick] - ref_c).max() < TOL
    assert hit >= 2


def test_random_csearch_vs_oracle(fc):
    """random_csearch (torsion_module.py:436-571) with the shuffle fixed: same kept sets, same
    coordinates, including the max_tries quirk (the stop is only evaluated on a kept set)"""
    base, tors, masks = _chain_case(28, 4, seed=41)
    tors5 = [tuple(t) + (n,) for t, n in zip(tors, (6, 3, 6, 2))]
    grid = o.cartesian_product((0, 60, 120, 180, 240, 300), (0, 120, 240), (0, 60, 120, 180, 240, 300), (0, 180))
    order = np.random.RandomState(5).permutation(len(grid))
    for n_out, max_tries, rotations in ((25, 10000, None), (1000, 37, None), (12, 10000, 2)):
        g = grid if rotations is None else grid[np.count_nonzero(grid, axis=1) == rotations]
        od = order[order < len(g)] if rotations is not None else order
        ref, ref_idx = o.random_csearch(base, tors, masks, g[od], n_out=n_out, max_tries=max_tries)
        out, idx = fc.torsion_module.random_csearch_core(base, tors5, masks, n_out=n_out, max_tries=max_tries,
                                                    rotations=rotations, order=od, return_indices=True)
        assert np.array_equal(idx, ref_idx) and len(idx) > 0
        assert np.abs(out - ref).max() < TOL
    # seeded shuffle = RandomState(seed).shuffle of the grid
    out = fc.torsion_module.random_csearch_core(base, tors5, masks, n_out=10, seed=3)
    g = grid.copy()
    np.random.RandomState(3).shuffle(g)
    ref, _ = o.random_csearch(base, tors, masks, g, n_out=10)
    assert np.abs(out - ref).max() < TOL


def test_align_by_moi_vs_oracle(fc):
    """align_by_moi (hypermolecule_class.py:45-86): centring in place + the literal
    get_alignment_matrix of the two diagonal moment arrays"""
    rng = np.random.default_rng(43)
    X = rng.normal(scale=2.5, size=(9, 21, 3)) + rng.normal(scale=4.0, size=(9, 1, 3))
    atoms = np.array(list("CHONCHHCCOHHNCCHHHOCH"))
    masses = np.array([fc.pt.pt.mass(a) for a in atoms])
    ref = o.align_by_moi(masses, X.copy())
    mine = X.copy()
    out = fc.hypermolecule_class.align_by_moi(atoms, mine)
    assert np.abs(out - ref).max() < TOL
    assert np.abs(mine.mean(axis=1)).max() < 1e-12  # centred in place like the reference
    assert np.abs(out[0] - (X[0] - X[0].mean(axis=0))).max() < 1e-12


def test_scan_with_fingerprints_equals_two_passes(fc):
    """fc_torsion_scan_fingerprints: fingerprints taken inside the scan kernel == fingerprints of
    the conformers the plain scan returns (and == the oracle's), with and without coordinates"""
    base, tors, masks = _chain_case(30, 4, seed=36)
    quads = np.array([[0, 3, 9, 20], [5, 6, 7, 8], [29, 2, 14, 1]] + [list(t) for t in tors])
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4)[::7]
    out, rot = fc.torsion_module.torsion_scan(base, tors, masks, angles)
    tf, rot2 = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, quads)
    tf3, rot3, out3 = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, quads, want_coords=True)
    assert np.array_equal(rot, rot2) and np.array_equal(rot, rot3) and np.array_equal(out, out3)
    assert np.array_equal(tf, fc.torsion_module.get_tf_mat(out, quads)) and np.array_equal(tf, tf3)
    assert np.abs(tf - o.get_tf_mat(o.torsion_scan(base, tors, masks, angles)[0], quads)).max() < TOL


@pytest.mark.parametrize("q", [2, 5, 11])
def test_tfd_first_match_two_phases(fc, monkeypatch, q):
    """fc_tfd_first_match with a bounded look-ahead per row block and the rows still open handed to the
    column-chunked kernel (the form long arrays take; forced here with a look-ahead of 64 ... 640 columns) ==
    the one-phase kernel == a NumPy first-match, on fingerprints with many unmatched rows and far matches"""
    from firecode_amd import _lib as L

    rng = np.random.default_rng(61 + q)
    n = 6000
    centres = rng.uniform(-180, 180, size=(700, q))
    tf = centres[rng.integers(0, len(centres), n)] + rng.normal(scale=1.5, size=(n, q))
    tf[rng.integers(0, n, 300)] = rng.uniform(-180, 180, size=(300, q))  # rows without any partner
    tf = (tf + 180) % 360 - 180
    ref = np.full(n, -1, dtype=np.int64)
    for i in range(n - 1):
        d = np.abs(tf[i + 1:] - tf[i])
        d = np.abs(d - (d > 180) * 360)
        hit = np.flatnonzero(d.sum(axis=1) < 10)
        if len(hit):
            ref[i] = i + 1 + hit[0]
    assert (ref < 0).sum() > 100 and ((ref - np.arange(n))[ref >= 0] > 1000).sum() > 50
    # angles at the wrap-around (+-180) as well: the window boxes and the deltas have to agree there
    assert (np.abs(tf) > 175).any()
    # (two phases: 16-bit angles -- a dense phase, then the ordered walk over windows -- or, FC_TFD_U16=0, the fp32
    # pre-filter kernels with column chunks)
    for u16 in (None, "0"):
        if u16 is None:
            monkeypatch.delenv("FC_TFD_U16", raising=False)
        else:
            monkeypatch.setenv("FC_TFD_U16", u16)
        for look in ("0", "64", "640", None):
            if look is None:
                monkeypatch.delenv("FC_TFD_LOOKAHEAD", raising=False)
            else:
                monkeypatch.setenv("FC_TFD_LOOKAHEAD", look)
            fm = np.zeros(n, dtype=np.int64)
            L.call("fc_tfd_first_match", L.pf(np.ascontiguousarray(tf)), n, q, 10.0, L.pi(fm))
            assert np.array_equal(fm, ref), (u16, look)


def _first_match_numpy(tf, thresh):
    n = len(tf)
    ref = np.full(n, -1, dtype=np.int64)
    for i in range(n - 1):
        d = np.abs(tf[i + 1:] - tf[i])
        d = np.abs(d - (d > 180) * 360)
        hit = np.flatnonzero(d.sum(axis=1) < thresh)
        if len(hit):
            ref[i] = i + 1 + hit[0]
    return ref


@pytest.mark.parametrize("q", [3, 8, 9])
def test_tfd_first_match_16bit_band_wrap_and_refusal(fc, monkeypatch, q):
    """The 16-bit first-match kernels (k_tfd_first_match_dense_u16 / _walk_u16) decide a pair from the sum of absolute
    differences of 16-bit angles only outside a band of +-10 units (0.055 deg) around the threshold: pairs planted AT
    the threshold (sum = 10 -+ 1e-9 ... 0.07) must come out as the fp64 sum in NumPy's order says, next to the dense
    phase (partners a few rows away), in the walk (thousands of rows away), across the +-180 wrap; fingerprints the
    16 bits cannot stand for (an angle beyond +-270, a NaN) are refused by the kernels and redone by the fp32 ones
    (reference: torsion_module.py:1056-1067 through `_first_match_numpy`)"""
    from firecode_amd import _lib as L

    rng = np.random.default_rng(900 + q)
    n = 5000
    tf = rng.uniform(-180, 180, size=(n, q))  # far from each other: no accidental partners
    eps = [0.0, 1e-9, -1e-9, 3e-3, -3e-3, 0.02, -0.02, 0.05, -0.05, 0.07, -0.07, 0.5, -0.5]
    rows = rng.choice(n - 4200, size=len(eps) * 3, replace=False)
    planted = []
    for k, i in enumerate(rows):
        e = eps[k % len(eps)]
        gap = (3, 700, 4100)[k // len(eps)]  # dense phase / a few windows on / many windows on
        w = rng.dirichlet(np.ones(q)) * (10.0 + e)
        sign = rng.choice([-1.0, 1.0], size=q)
        j = i + gap
        tf[j] = tf[i] + sign * w
        planted.append((i, j))
    # two planted pairs across the wrap
    tf[100] = 179.0
    tf[4000] = -179.5
    tf[4000, 0] = -178.0  # sum = 1.5 (q - 1) + 3.0 < 10 only for small q: both outcomes occur over the parametrisation
    tf = (tf + 180) % 360 - 180
    ref = _first_match_numpy(tf, 10.0)
    assert sum(ref[i] == j for i, j in planted) >= len(planted) // 3 and sum(ref[i] != j for i, j in planted) >= len(planted) // 4
    for look in ("32", "512", None):  # (None: one phase at this size -- the fp32 kernel)
        if look is None:
            monkeypatch.delenv("FC_TFD_LOOKAHEAD", raising=False)
        else:
            monkeypatch.setenv("FC_TFD_LOOKAHEAD", look)
        fm = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(np.ascontiguousarray(tf)), n, q, 10.0, L.pi(fm))
        assert np.array_equal(fm, ref), look
    # refusal: the reference's delta is not the circular distance beyond |a - b| = 540, and NaN is similar to nothing
    monkeypatch.setenv("FC_TFD_LOOKAHEAD", "64")
    for bad in (300.0, -512.0, np.nan):
        tf2 = tf.copy()
        tf2[1234, q - 1] = bad
        tf2[10, 0] = 271.0 if np.isnan(bad) else tf2[10, 0]
        with np.errstate(invalid="ignore"):
            ref2 = _first_match_numpy(tf2, 10.0)
        fm = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(np.ascontiguousarray(tf2)), n, q, 10.0, L.pi(fm))
        assert np.array_equal(fm, ref2), bad
    # thresholds beside the usual one, zero and negative included (nothing is similar)
    for thr in (0.0, -1.0, 0.3, 40.0):
        refT = _first_match_numpy(tf, thr)
        fm = np.zeros(n, dtype=np.int64)
        L.call("fc_tfd_first_match", L.pf(np.ascontiguousarray(tf)), n, q, float(thr), L.pi(fm))
        assert np.array_equal(fm, refT), thr


def test_scan_tfd_fused_equals_scan_then_prune(fc):
    """fc_torsion_scan_tfd (fingerprints resident on the device between the scan and the TFD prune) == the two
    calls with the fingerprints through the host, and == the oracle's literal loop on the same rows"""
    for n_tors, stride, seed in ((4, 1, 37), (5, 3, 38)):
        base, tors, masks = _chain_case(32, n_tors, seed=seed)
        angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * n_tors)[::stride]
        tf, rot = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, tors)
        kept = np.flatnonzero(rot != 0)
        tf_all = np.concatenate([fc.torsion_module.get_torsion_fingerprint(base, tors)[None], tf[kept]])
        mask = fc.torsion_module.prune_tfd_from_tf_mat(tf_all, 10)
        rot2, keep = fc.torsion_module.torsion_scan_tfd(base, tors, masks, angles, tors, tfd_thresh=10)
        assert np.array_equal(rot, rot2)
        expect = np.zeros(len(angles) + 1, dtype=bool)
        expect[0] = mask[0]
        expect[1 + kept] = mask[1:]
        assert np.array_equal(keep, expect) and 0 < keep.sum() < len(keep)
        if len(tf_all) <= 2000:
            assert np.array_equal(mask, o.prune_tfd_from_tf_mat(tf_all, 10))
    # nothing rotates (every angle zero): the starting structure alone
    rot0, keep0 = fc.torsion_module.torsion_scan_tfd(base, tors, masks, np.zeros((3, n_tors), dtype=np.int64), tors)
    assert not rot0.any() and keep0.tolist() == [True, False, False, False]


def test_scan_tfd_over_the_device_generated_grid(fc):
    """fc_torsion_scan_tfd_grid (the n-fold grid generated on the device in cartesian_product's row order) == the same
    call on the host-built grid, mixed n-folds included; cartesian_rows_at gives the rows back; clustered_csearch
    (which uses both) returns what the scan of the host-built grid returns"""
    for n_tors, folds, seed in ((4, (6, 6, 6, 6), 41), (5, (6, 2, 3, 4, 6), 42), (1, (6,), 43), (2, (3, 4), 44)):
        base, tors, masks = _chain_case(32, n_tors, seed=seed)
        values = [fc.torsion_module.N_FOLD_ANGLES[f] for f in folds]
        angles = o.cartesian_product(*values)
        assert np.array_equal(fc.utils.cartesian_rows_at(values, np.arange(len(angles))), angles)
        rot, keep = fc.torsion_module.torsion_scan_tfd(base, tors, masks, angles, tors, tfd_thresh=10)
        rot_g, keep_g = fc.torsion_module.torsion_scan_tfd_grid(base, tors, masks, values, tors, tfd_thresh=10)
        assert np.array_equal(rot, rot_g) and np.array_equal(keep, keep_g) and keep.sum() >= 1
    base, tors, masks = _chain_case(32, 4, seed=41)
    t5 = [tuple(int(v) for v in t) + (6,) for t in tors]
    got = fc.torsion_module.clustered_csearch_core(base, t5, masks, n_out=10 ** 6)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 4)
    rot, keep = fc.torsion_module.torsion_scan_tfd(base, tors, masks, angles, tors, tfd_thresh=10)
    want = fc.torsion_module.torsion_scan(base, tors, masks, angles[np.flatnonzero(keep[1:])])[0]
    if keep[0]:
        want = np.concatenate([base[None], want])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_clash_functions_on_the_reference_fixture_molecules(fc, golden):
    """count_clashes / fragment compenetration_check on the molecules of the reference's own test
    files, against the reference's own outputs"""
    for name in golden["fx_names"]:
        coords = golden[f"fx_{name}_coords"]
        A = coords.shape[1]
        assert np.array_equal(fc.algebra.count_clashes_batch(coords), golden[f"fx_{name}_clashes"])
        for k, mc in enumerate((0, 1, 2, 4, 8)):
            out = fc.utils.compenetration_check_batch(coords, ids=[A // 2, A - A // 2], thresh=1.6, max_clashes=mc)
            assert np.array_equal(out, golden[f"fx_{name}_frag"][:, k])


def test_rotate_dihedral_and_comp_check(fc, golden):
    base, tors, masks = _chain_case(20, 2, seed=32)
    new = fc.utils.rotate_dihedral(base, tors[0], 120, mask=masks[0])
    assert np.abs(new - o.rotate_dihedral(base, tors[0], 120, masks[0])).max() < TOL
    out = [fc.torsion_module.torsion_comp_check(c, tuple(t), m.copy()) for c, t, m in
           zip(golden["tc_in"], golden["tc_tors"], golden["tc_mask"])]
    assert np.array_equal(out, golden["tc_out"])
    out = [fc.torsion_module.torsion_comp_check(c, tuple(t), m.copy(), max_clashes=2) for c, t, m in
           zip(golden["tc_in"][:20], golden["tc_tors"][:20], golden["tc_mask"][:20])]
    assert np.array_equal(out, golden["tc_out_mc2"][:20])


def test_fingerprints_and_tfd(fc, golden):
    rng = np.random.default_rng(33)
    X = rng.normal(scale=2.0, size=(50, 16, 3))
    quads = np.array([rng.choice(16, 4, replace=False) for _ in range(7)])
    tf = fc.torsion_module.get_tf_mat(X, quads)
    tf0 = o.get_tf_mat(X, quads)
    assert np.abs(tf - tf0).max() < TOL
    assert abs(fc.algebra.dihedral(X[0][quads[0]]) - tf0[0, 0]) < TOL
    out = [fc.torsion_module.tfd_similarity(a, b) for a, b in zip(golden["tfd_a"], golden["tfd_b"])]
    assert np.array_equal(out, golden["tfd_out"])


def test_tfd_long_fingerprints_numpy_sum_order(fc, golden):
    """Q >= 8: np.sum's 8-lane pairwise order decides sums that straddle the threshold"""
    a, b = golden["tfdl_a"], golden["tfdl_b"]
    out = [fc.torsion_module.tfd_similarity(x, y) for x, y in zip(a[:120], b[:120])]
    assert np.array_equal(out, golden["tfdl_out"][:120])
    assert 20 < golden["tfdl_out"][:120].sum() < 100


@pytest.mark.parametrize("name", ["tfdp_small", "tfdp_mid", "tfdp_big", "tfdp_dense", "tfdp_q8", "tfdp_q11", "tfdp_q19"])
def test_tfd_bits_and_prune_golden(fc, golden, name, monkeypatch):
    """TFD similarity bits on the GPU + the reference's bookkeeping = the
    mask the reference's own prune_conformers_tfd produced."""
    tf = golden[name + "_tf"]
    n = len(tf)
    from firecode_amd import torsion_module as tm

    monkeypatch.setattr(tm, "get_tf_mat", lambda s, q: tf)
    _, mask = tm.prune_conformers_tfd(np.zeros((n, 4, 3)), np.zeros((tf.shape[1], 4), dtype=int), thresh=10)
    assert np.array_equal(mask, golden[name + "_mask"])


# ---------------------------------------------------------------- Ensemble driver
def test_ensemble_similarity_pruning_cfg1(fc, tmp_path):
    """BASELINE config 1: 200 x 30 written to .xyz (8 decimals), re-read, pruned."""
    X, atoms, asg = syn.synthetic_ensemble(200, 30, seed=1)
    ens = fc.ensemble.Ensemble(atoms=atoms, coords=X, basename="cfg1", logfunction=None)
    path = tmp_path / "cfg1.xyz"
    ens.to_xyz(path)
    assert path.read_text() == o.ensemble_to_xyz_text(atoms, X, "cfg1")
    back = fc.ensemble.Ensemble.from_xyz(path)
    back.logfunction = None
    Xr = back.coords.copy()
    back.similarity_pruning(moi=False, rmsd=True, max_rmsd=0.5)
    _, ref = o.prune_by_rmsd(Xr, atoms, 0.5)
    assert np.array_equal(back.coords, Xr[ref])
    assert len(back.coords) == len(np.unique(asg))


# ---------------------------------------------------------------- sharded ladder on one GPU
@pytest.mark.parametrize("world,row_block", [(2, 128), (3, 128), (4, 256)])
def test_sharded_prune_all_ranks_on_one_gpu(fc, world, row_block):
    """every 'rank' is a DeviceEnsemble on this GPU; the all-gather is an
    in-process stack -- exercises fc_prune_rmsd_begin / fc_prune_level and the
    snake row ownership on the real kernels"""
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1100, 24, seed=50 + world)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    ranks = [fc.DeviceEnsemble(X, center=True) for _ in range(world)]
    pairs = 0
    for r, ens in enumerate(ranks):
        stats = ens.prune_begin(0.5, 1.0, r, world, row_block=row_block)
        pairs += int(stats[0])
    assert pairs == len(X) * (len(X) - 1) // 2  # every pair has exactly one owner
    mask = np.ones(len(X), dtype=np.uint8)
    for k in fdist.LADDER:
        if k == 1 or 20 * k < int(mask.sum()):
            mask = np.stack([ens.prune_level(k, mask) for ens in ranks]).min(axis=0)
    for ens in ranks:
        ens.close()
    assert np.array_equal(mask.astype(bool), ref)


# ---------------------------------------------------------------- a14: bimolecular embed grid
def _embed_case(seed, n1=3, n2=4, A1=14, A2=11, nr1=2, nr2=1):
    rng = np.random.default_rng(seed)
    m1 = rng.normal(scale=1.8, size=(n1, A1, 3))
    m2 = rng.normal(scale=1.8, size=(n2, A2, 3))
    r1 = rng.choice(A1, nr1, replace=False)
    r2 = rng.choice(A2, nr2, replace=False)
    # pivots: two pseudo-orbital centres near the reactive atoms, pushed outwards
    def pivots(m, r):
        out = np.empty((len(m), 2, 3))
        for c, x in enumerate(m):
            ra = x[r[0]]
            rb = x[r[-1]]
            out[c, 0] = ra * 1.6 + rng.normal(scale=0.3, size=3)
            out[c, 1] = rb * 1.6 + rng.normal(scale=0.3, size=3) + (0 if len(r) == 2 else 1.2)
        return out
    return m1, r1, pivots(m1, r1), m2, r2, pivots(m2, r2)


@pytest.mark.parametrize("seed,nr1,nr2", [(61, 2, 1), (62, 1, 1), (63, 2, 2)])
def test_embed_grid_vs_oracle(fc, seed, nr1, nr2):
    m1, r1, pv1, m2, r2, pv2 = _embed_case(seed, nr1=nr1, nr2=nr2)
    steps, rr = 3, 45.0
    angles = np.arange(steps + 1) * 2 * rr / steps - rr
    R1, t1 = fc.embeds.embed_mol_transforms(m1, r1, pv1, 0, angles)
    R2, t2 = fc.embeds.embed_mol_transforms(m2, r2, pv2, 1, angles)
    ok, counts, ms = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=1.5,
                                                max_clashes=0, return_counts=True)
    assert ok.shape == (len(m2), len(m1), 2, len(angles), len(angles))
    # the reference's loop order: conf pairs (second molecule slowest), orientation, angle pairs
    conf_ids = o.cartesian_product(range(len(m1)), range(len(m2)))
    ang_ids = o.cartesian_product(range(steps + 1), range(steps + 1))
    ref_ok = np.zeros_like(ok)
    ref_cnt = np.zeros(ok.shape, dtype=np.int64)
    worst_R = worst_t = 0.0
    for c1, c2 in conf_ids:
        for ori in (0, 1):
            for i1, i2 in ang_ids:
                Ra, ta, Rb, tb = o.bimol_pose_transforms(m1[c1], m2[c2], r1, r2, pv1[c1], pv2[c2],
                                                         (angles[i1], angles[i2]), ori)
                worst_R = max(worst_R, np.abs(Ra - R1[c1, ori, i1]).max(), np.abs(Rb - R2[c2, ori, i2]).max())
                worst_t = max(worst_t, np.abs(ta - t1[c1, ori, i1]).max(), np.abs(tb - t2[c2, ori, i2]).max())
                pose = o.get_embed([m1[c1], m2[c2]], [Ra, Rb], [ta, tb])
                ref_ok[c2, c1, ori, i2, i1] = o.compenetration_check(pose, ids=[m1.shape[1], m2.shape[1]],
                                                                     thresh=1.5)
                from scipy.spatial.distance import cdist
                ref_cnt[c2, c1, ori, i2, i1] = np.count_nonzero(cdist(pose[m1.shape[1]:], pose[:m1.shape[1]]) < 1.5)
    assert worst_R < TOL and worst_t < TOL
    assert np.array_equal(ok, ref_ok)
    assert 0 < ok.sum() < ok.size
    # flattened index == reference iteration order
    flat = ok.reshape(-1)
    k = 0
    for c1, c2 in conf_ids[:3]:
        for ori in (0, 1):
            for i1, i2 in ang_ids:
                assert flat[((c2 * len(m1) + c1) * 2 + ori) * len(ang_ids) + (i2 * (steps + 1) + i1)] == \
                    ref_ok[c2, c1, ori, i2, i1]
    ok3, _ = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=1.5, max_clashes=3)
    assert np.array_equal(ok3, ref_cnt <= 3) and ok3.sum() > ok.sum()
    assert np.array_equal(np.minimum(counts, 1), np.minimum(ref_cnt, 1))  # counts saturate past max_clashes


@pytest.mark.parametrize("a1,a2,seed", [(13, 11, 71), (24, 17, 72), (9, 40, 73)])
def test_embed_grid_fp32_screen_equals_all_fp64_kernel(fc, monkeypatch, a1, a2, seed):
    """the pose-grid clash kernel rules pairs out in packed fp32 and recounts the rest exactly;
    pass flags AND counts must equal the all-fp64 kernel (FC_GRID_F64=1) bit for bit: odd atom
    counts (padding atom, partial atom groups), thresholds that sit exactly ON an interatomic
    distance of a pose (the fp32 screen cannot decide, the exact recount says "no clash" and the
    lane has to resume), tiny and huge thresholds, several max_clashes"""
    rng = np.random.default_rng(seed)
    m1, r1, pv1, m2, r2, pv2 = _embed_case(seed, nr1=2, nr2=1)
    m1, m2 = m1[:, :a1].copy(), np.concatenate([m2] * 2, axis=1)[:, :a2].copy()
    m2[:, m2.shape[1] // 2:] += rng.normal(scale=0.7, size=m2[:, m2.shape[1] // 2:].shape)
    r1, r2 = np.array([0, 1]), np.array([2])
    angles = np.array([-45.0, 0.0, 30.0])

    def both(thresh, mc):
        monkeypatch.setenv("FC_GRID_F64", "0")
        ok, cnt, _ = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=thresh, max_clashes=mc,
                                                return_counts=True)
        monkeypatch.setenv("FC_GRID_F64", "1")
        ok64, cnt64, _ = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=thresh, max_clashes=mc,
                                                    return_counts=True)
        assert np.array_equal(ok, ok64) and np.array_equal(cnt, cnt64)
        return ok, cnt

    # distances of one real pose: thresholds exactly on them
    R1, t1 = fc.embeds.embed_mol_transforms(m1, r1, pv1, 0, angles)
    R2, t2 = fc.embeds.embed_mol_transforms(m2, r2, pv2, 1, angles)
    pose = o.get_embed([m1[0], m2[0]], [R1[0, 0, 1], R2[0, 0, 2]], [t1[0, 0, 1], t2[0, 0, 2]])
    from scipy.spatial.distance import cdist

    d = np.sort(cdist(pose[a1:], pose[:a1]).reshape(-1))
    seen = []
    for thresh in [1.5, 1e-3, 0.4, 40.0, float(d[0]), float(d[3]), float(np.nextafter(d[3], 9.0)), float(d[len(d) // 2])]:
        for mc in (0, 1, 2):
            ok, _ = both(thresh, mc)
            seen.append(ok.mean())
    assert min(seen) == 0.0 and max(seen) == 1.0 and len(set(seen)) > 4  # all regimes were visited
    monkeypatch.delenv("FC_GRID_F64")


# ---------------------------------------------------------------- graph clash, fitness, drivers
def test_compenetration_graph_mode_golden(fc, golden):
    edges = golden["cp_graph_edges"]
    for mc in (0, 2):
        out = fc.utils.compenetration_check_batch(golden["cpg_in"], graph=edges, thresh=1.2, max_clashes=mc)
        assert np.array_equal(out, golden[f"cp_graph_{mc}"])
    import networkx as nx

    g = nx.Graph([tuple(e) for e in edges])
    assert fc.utils.compenetration_check(golden["cpg_in"][5], graph=g, thresh=1.2) == bool(golden["cp_graph_0"][5])


def test_fitness_check(fc):
    rng = np.random.default_rng(71)
    X = rng.normal(scale=2.0, size=(40, 12, 3))
    cons = np.array([[0, 5], [3, 7], [2, 11]])
    targets = [2.0, None, 1.5]
    ok, err = fc.utils.fitness_check_batch(X, cons, [targets], threshold=3.0)
    ref = np.array([o.fitness_check(x, cons, targets, 3.0) for x in X])
    assert np.array_equal(ok, ref) and 0 < ok.sum() < len(ok)
    assert fc.utils.fitness_check(X[0], cons, targets, 3.0) == bool(ref[0])
    m = fc.refining.fitness_refining(X, cons, [targets], threshold=3.0)
    assert np.array_equal(m, ref)


def test_refining_drivers(fc, golden):
    X, atoms, asg = syn.synthetic_ensemble(400, 18, seed=72)
    logs = []
    mask = fc.refining.similarity_refining(X, atoms, rmsd_thr=0.5, moi=True, rmsd=True, logfunction=logs.append)
    _, m1 = o.prune_by_moment_of_inertia(X, atoms)
    _, m2 = o.prune_by_rmsd(X[m1], atoms, 0.5)
    ref = np.zeros(len(X), dtype=bool)
    ref[np.flatnonzero(m1)[m2]] = True
    assert np.array_equal(mask, ref)
    assert any("RMSD similarity" in s or "MOI similarity" in s for s in logs)
    cm = fc.refining.compenetration_refining(golden["cp_in"], ids=[20, 16], clash_thresh=1.5, max_clashes=0)
    assert np.array_equal(cm, golden["cp_bi_1.5_0"])
    em = fc.refining.energy_pruning(golden["enp_energies"], 10.0)
    assert em.sum() == int(golden["enp_kept10"])
    assert fc.refining.dynamic_energy_thr(golden["enp_energies"], 0.5) == float(golden["enp_thr0p5"])


def test_clustered_csearch_driver(fc):
    base, tors, masks = _chain_case(26, 3, seed=73)
    torsions = [tuple(t) + (6,) for t in tors]
    out = fc.torsion_module.clustered_csearch_core(base, torsions, masks, n_out=10_000)
    # oracle pipeline: scan -> keep rotated -> TFD prune
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 3)
    sc, rot = o.torsion_scan(base, tors, masks, angles)
    new = np.concatenate([base[None], sc[rot != 0]])
    ref, _ = o.prune_conformers_tfd(new, tors)
    assert out.shape == ref.shape and np.abs(out - ref).max() < TOL
    sub = fc.torsion_module.most_diverse_conformers(5, list(ref), seed=1)
    assert len(sub) == 5


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pairs_exchange_on_one_gpu(fc, world):
    """the one-all-gather path: every rank contributes its similar pairs, every
    rank replays the ladder from the union"""
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(900, 20, seed=80 + world)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    ranks = [fc.DeviceEnsemble(X, center=True) for _ in range(world)]
    lists = []
    for r, ens in enumerate(ranks):
        ens.prune_begin(0.5, 1.0, r, world, row_block=128)
        lists.append(ens.similar_pairs())
    assert sum(len(l) for l in lists) == np.triu(S0, 1).sum()
    union = np.concatenate(lists + [np.full(7, fdist.PAD, dtype=np.uint64)])  # padding is ignored
    for ens in ranks:
        assert np.array_equal(ens.prune_from_pairs(union), ref)
        ens.close()


def test_sharded_prune_behind_the_c_abi_single_rank(fc, monkeypatch):
    """fc_prune_rmsd_sharded / fc_bench_prune_rmsd_sharded with the real RCCL communicator of one
    rank (fc_comm_init through firecode_amd.dist.comm_init_from_env -- no torch): stream ordering
    between the library's kernels and RCCL's stream, no host wait in between; then the dense
    fallback (4-entry candidate queue -> one fc_allgather_mask per ladder level)"""
    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1500, 30, seed=195)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FC_COMM_ID"):
        monkeypatch.delenv(k, raising=False)
    assert fdist.comm_init_from_env() == (0, 1, 0)
    try:
        assert _lib.comm_info() == (0, 1)
        assert np.array_equal(_lib.allgather_mask(np.arange(7, dtype=np.uint8)), np.arange(7, dtype=np.uint8)[None])
        _lib.comm_barrier()
        with fc.DeviceEnsemble(X, center=True) as ens:
            for _ in range(3):
                mask, stats = fdist.prune_by_rmsd_sharded_rccl(ens, 0.5)
                assert np.array_equal(mask, ref) and stats[2] == np.triu(S0, 1).sum() and stats[5] == ref.sum()
                assert stats[0] == len(X) * (len(X) - 1) // 2 and stats[4] > 0
            for overlap in (False, True):
                tk, ts, mask, stats = ens.bench_prune_sharded(0.5, 1.0, reps=5, overlap=overlap)
                assert np.array_equal(mask, ref) and stats[5] == ref.sum() and tk > 0 and ts > 0
            mask, _ = ens.prune(0.5, 1.0)  # the plain call still works on the same ensemble afterwards
            assert np.array_equal(mask, ref)
        monkeypatch.setenv("FC_PAIRQ_CAP", "4")
        with fc.DeviceEnsemble(X, center=True) as ens:
            mask, stats = ens.prune_sharded(0.5, 1.0)
            assert np.array_equal(mask, ref) and stats[5] == ref.sum()
    finally:
        _lib.comm_destroy()
    assert _lib.comm_info() == (0, 1)


@pytest.mark.parametrize("world,overlap", [(2, False), (2, True), (3, True)])
def test_sharded_prune_c_path_logical_ranks(fc, world, overlap):
    """The C pipeline with more than one rank, on one GPU: fc_debug_comm_loopback(rank, world) makes
    the library act as `rank` of `world` with the all-gather writing only its own slot of the
    workspace's receive buffer -- the ranks run one after the other on the same ensemble, the last
    one sees every message.  Same mask as the oracle, owned pairs add up."""
    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1100, 18, seed=199 + world)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    owner = fdist.owner_of_rows(len(X), world, 128)
    owned_total = 0
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            for r in list(range(1, world)) + [0]:
                _lib.call("fc_debug_comm_loopback", r, world)
                if overlap:
                    _, _, mask, stats = ens.bench_prune_sharded(0.5, 1.0, reps=4, overlap=True)
                else:
                    mask, stats = ens.prune_sharded(0.5, 1.0)
                assert stats[2] == int(np.triu(S0, 1)[owner == r].sum())
                owned_total += int(stats[0])
            assert np.array_equal(mask, ref) and stats[5] == ref.sum()
            assert owned_total == len(X) * (len(X) - 1) // 2
    finally:
        _lib.call("fc_debug_comm_loopback", -1, 0)


def _tri_objects(mols):
    from oracle import cyclical_ref as cy

    return [cy.Mol(m["coords"], m["reactive_indices"], [[cy.Pivot(*p) for p in pl] for pl in m["pivots"]],
                   m["reactive_cumnums"]) for m in mols]


@pytest.mark.parametrize("seed,thresh,pairing", [(2, 1.5, False), (5, 0.9, False), (7, 1.1, True)])
def test_trimolecular_cyclical_embed_vs_oracle(fc, seed, thresh, pairing):
    """cyclical_embed for three molecules (embeds.py:409-585): adjusted directions, clash pass
    flags, the sequential accept filter, the poses and the constrained indices, group by group"""
    from oracle import cyclical_ref as cy

    mols = syn.synthetic_trimolecular(n_conf=(2, 1, 2), n_atoms=(9, 12, 8), seed=seed, pivots_per_conf=(1, 2, 1))
    angles = o.cartesian_product(*[range(6)] * 3) * 2 * 45 / 5 - 45
    table = None
    if pairing:  # a pairing only some orientations realise
        cum = [list(m["reactive_cumnums"].values()) for m in mols]
        table = {"a": tuple(sorted((cum[0][1], cum[1][0])))}
    trace = []
    ref_poses, ref_ci = cy.cyclical_embed_trimolecular(_tri_objects(mols), angles, pairings_table=table,
                                                       clash_thresh=thresh, trace=trace)
    poses, ci, det = fc.embeds.cyclical_embed_trimolecular(mols, angles, pairings_table=table, clash_thresh=thresh,
                                                          return_details=True)
    groups = {(g[0], g[1], g[2]): g for g in trace}
    n_run = 0
    for j, (conf_ids, piv_ids) in enumerate(det["jobs"]):
        for v in range(8):
            key = (conf_ids, piv_ids, v)
            assert det["run"][j, v] == (key in groups)
            if key not in groups:
                assert not det["accepted"][j, v].any()
                continue
            n_run += 1
            _, _, _, d_ref, p_ref, a_ref = groups[key]
            assert np.abs(det["directions"][j, v] - d_ref).max() < TOL
            assert np.array_equal(det["passed"][j, v], p_ref)
            assert np.array_equal(det["accepted"][j, v], a_ref)
    assert n_run == len(trace) and n_run > 0
    if pairing:
        assert 0 < n_run < 8 * len(det["jobs"])
    assert poses.shape == ref_poses.shape and len(poses) > 0
    assert np.abs(poses - ref_poses).max() < TOL
    assert np.array_equal(ci, ref_ci)
    if thresh < 1.0:  # the accept filter had something to reject
        assert det["passed"].sum() > det["accepted"].sum()


def _tbu_ensemble(seed=0, n_backbone=4, jitter=0.01):
    """A chain C0-C1-C2-C3 whose C3 carries three methyls (a 3-fold locally symmetric rotor)
    and C0 carries two equivalent carbons (a 2-fold one); conformers = backbone dihedral in
    `n_backbone` values x rotor turned by 0/120/240 (+ a few degrees) x flipper by 0/180"""
    import networkx as nx

    rng = np.random.default_rng(seed)
    s, c = np.sin(np.radians(70.5)), np.cos(np.radians(70.5))
    heavy = [[-3.7, 1.4, 0.3], [-2.2, 1.3, 0.0], [-1.5, 0.0, 0.0], [0.0, 0.0, 0.0]]
    heavy += [[1.5 * c, 1.5 * s * np.cos(p), 1.5 * s * np.sin(p)] for p in np.radians([10.0, 130.0, 250.0])]
    d = np.array(heavy[0]) - np.array(heavy[1])
    d /= np.linalg.norm(d)
    perp = np.cross(d, [0.0, 0.0, 1.0])
    perp /= np.linalg.norm(perp)
    heavy += [list(np.array(heavy[0]) + 1.4 * (0.5 * d + 0.87 * perp)), list(np.array(heavy[0]) + 1.4 * (0.5 * d - 0.87 * perp))]
    # one hydrogen per methyl, on the C3->methyl line: the rotor stays 3-fold symmetric with its hydrogens
    hyd = [list(np.array(heavy[k]) * (1.0 + 1.09 / 1.5)) for k in (4, 5, 6)]
    base = np.array(heavy + hyd)
    atoms = np.array(["C"] * 9 + ["H"] * 3)
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (3, 5), (3, 6), (0, 7), (0, 8), (4, 9), (5, 10), (6, 11)]
    graph = nx.Graph(edges)
    nx.set_node_attributes(graph, {i: str(a) for i, a in enumerate(atoms)}, "atoms")  # as graphize() leaves them
    torsions = [(1, 2, 3, 4, 3), (2, 1, 0, 7, 2)]
    masks = np.array([fc_rotation_mask(graph, t[:4], len(atoms)) for t in torsions])
    backbone = (0, 1, 2, 3)
    bb_mask = fc_rotation_mask(graph, backbone, len(atoms))
    out = []
    for b in range(n_backbone):
        x0 = o.rotate_dihedral(base, backbone, 50.0 * b, bb_mask)
        for turn in (0.0, 120.0, 240.0):
            for flip in (0.0, 180.0):
                x = o.rotate_dihedral(x0, torsions[0][:4], turn + rng.uniform(-2, 2), masks[0])
                x = o.rotate_dihedral(x, torsions[1][:4], flip + rng.uniform(-2, 2), masks[1])
                x = x + rng.normal(scale=jitter, size=x.shape)
                out.append(x @ syn.random_rotation(rng).T + rng.normal(scale=3.0, size=3))
    order = rng.permutation(len(out))
    return np.array(out)[order], atoms, graph, torsions, masks


def fc_rotation_mask(graph, torsion, n):
    from firecode_amd.pruner import rotation_mask

    return rotation_mask(graph, torsion, n)


def test_prune_by_rmsd_rot_corr_vs_oracle(fc):
    """a7 (PARITY UNPINNED restatement): similarity bits, mask, energy window; and the point of
    the function -- rotamers of locally symmetric groups collapse, which plain RMSD keeps apart"""
    X, atoms, graph, torsions, masks = _tbu_ensemble(seed=3)
    angle_sets = [(0, 120, 240), (0, 180)]
    quads = [t[:4] for t in torsions]
    S0 = o.prune_by_rmsd_rot_corr(X, atoms, quads, masks, angle_sets, max_rmsd=0.25, return_matrix=True)
    _, ref_mask = o.prune_by_rmsd_rot_corr(X, atoms, quads, masks, angle_sets, max_rmsd=0.25)
    kept, mask, bits = fc.pruner.prune_by_rmsd_rot_corr(X, atoms, graph, max_rmsd=0.25, torsions=torsions,
                                                        return_bits=True)
    from firecode_amd._lib import unpack_bits

    assert np.array_equal(unpack_bits(bits, len(X)), S0)
    assert np.array_equal(mask, ref_mask) and np.array_equal(kept, X[mask])
    assert mask.sum() == 4  # one per backbone dihedral
    _, plain = fc.pruner.prune_by_rmsd(X, atoms, max_rmsd=0.25)
    assert plain.sum() > mask.sum()
    # masks given explicitly == derived from the graph; no torsions == plain prune of centred structures
    _, mask2 = fc.pruner.prune_by_rmsd_rot_corr(X, atoms, None, max_rmsd=0.25, torsions=torsions, rotation_masks=masks)
    assert np.array_equal(mask2, mask)
    _, mask3 = fc.pruner.prune_by_rmsd_rot_corr(X, atoms, None, max_rmsd=0.25, torsions=[])
    _, ref3 = o.prune_by_rmsd_rot_corr(X, atoms, [], [], [], max_rmsd=0.25)
    assert np.array_equal(mask3, ref3)
    # THE REFERENCE'S CALL (firecode/ensemble.py:253-260): structures, atoms, graph, keywords -- the
    # locally symmetric torsions are perceived from the graph, the rotamers collapse all the same
    logged = []
    kept4, mask4 = fc.pruner.prune_by_rmsd_rot_corr(X, atoms, graph, max_rmsd=0.25, energies=None, max_dE=1.0,
                                                    logfunction=logged.append, debugfunction=logged.append)
    assert np.array_equal(mask4, mask) and np.array_equal(kept4, X[mask]) and "2 symmetric torsions" in logged[0]
    with pytest.raises(ValueError):  # never an uncorrected prune under this name
        fc.pruner.prune_by_rmsd_rot_corr(X, atoms, max_rmsd=0.25)
    # energy window, processing order by energy
    en = np.random.default_rng(4).uniform(0, 3, size=len(X))
    _, ref_e = o.prune_by_rmsd_rot_corr(X, atoms, quads, masks, angle_sets, max_rmsd=0.25, energies=en, max_dE=1.0)
    _, mask_e = fc.pruner.prune_by_rmsd_rot_corr(X, atoms, graph, max_rmsd=0.25, torsions=torsions, energies=en,
                                                 max_dE=1.0)
    assert np.array_equal(mask_e, ref_e) and mask_e.sum() >= mask.sum()
    # the driver (ensemble.py:185-276) with the third stage switched on
    lines = []
    ens = fc.ensemble.Ensemble(atoms=atoms, coords=X.copy(), basename="rotor", logfunction=lines.append)
    ens.similarity_pruning(moi=False, rmsd=True, rmsd_rot_corr=True, symmetric_torsions=torsions, graph=graph)
    assert len(ens.coords) == 4 and any("symmetry-corrected RMSD" in ln for ln in lines)
    ens = fc.ensemble.Ensemble(atoms=atoms, coords=X.copy(), basename="rotor", logfunction=lines.append)
    ens.similarity_pruning(moi=False, rmsd=True, rmsd_rot_corr=True, graph=graph)  # perception from the graph
    assert len(ens.coords) == 4
    m = fc.refining.similarity_refining(X, atoms, rmsd_thr=0.25, moi=False, rmsd_rot_corr=True,
                                        symmetric_torsions=torsions, graph=graph)
    assert m.sum() == 4


def test_sharded_scan_and_pose_grid_logical_ranks(fc):
    """SURVEY 8e item 3 with the real kernels: 3 logical ranks on one GPU, the all-gather
    emulated by stacking the ranks' packed masks"""
    from firecode_amd import dist as fdist

    base, tors, masks = _chain_case(24, 3, seed=51)
    angles = o.cartesian_product((0, 120, 240), (0, 60, 120, 180, 240, 300), (0, 180))
    full_out, full_rot = fc.torsion_module.torsion_scan(base, tors, masks, angles)
    rng = np.random.default_rng(52)
    m1 = rng.normal(scale=1.8, size=(4, 9, 3))
    m2 = rng.normal(scale=1.8, size=(7, 8, 3))
    r1, r2 = np.array([0, 4]), np.array([2, 5])
    p1 = np.stack([m1[:, 0] + 0.9, m1[:, 4] - 0.8], axis=1)
    p2 = np.stack([m2[:, 2] + 0.7, m2[:, 5] - 1.0], axis=1)
    ang = fc.host_helpers.systematic_angles(5, 45)
    full_ok, _ = fc.embeds.embed_grid_clash(m1, r1, p1, m2, r2, p2, ang, thresh=1.3)
    world = 3
    sent_scan, sent_grid = [], []
    for r in range(world):  # first pass: what every rank would send
        fdist.torsion_scan_sharded(base, tors, masks, angles, rank=r, world=world,
                                   allgather_fn=lambda p: sent_scan.append(p) or np.zeros((world, len(p)), np.uint8))
        fdist.embed_grid_clash_sharded(m1, r1, p1, m2, r2, p2, ang, rank=r, world=world, thresh=1.3,
                                       allgather_fn=lambda p: sent_grid.append(p) or np.zeros((world, len(p)), np.uint8))
    for r in range(world):  # second pass: with the gathered messages
        out, rot, (lo, hi), keep = fdist.torsion_scan_sharded(base, tors, masks, angles, rank=r, world=world,
                                                              allgather_fn=lambda p: np.stack(sent_scan))
        assert np.array_equal(rot, full_rot[lo:hi]) and np.array_equal(out, full_out[lo:hi])
        assert np.array_equal(keep, full_rot != 0)
        ok = fdist.embed_grid_clash_sharded(m1, r1, p1, m2, r2, p2, ang, rank=r, world=world, thresh=1.3,
                                            allgather_fn=lambda p: np.stack(sent_grid))
        assert np.array_equal(ok, full_ok)
    assert 0 < full_ok.sum() < full_ok.size


# ---------------------------------------------------------------- screen-kernel variants / odd shapes
@pytest.mark.parametrize("n,a", [(2, 1), (3, 2), (70, 3), (130, 5), (200, 80), (150, 104), (140, 110), (90, 130), (150, 160), (130, 192), (100, 193),
                                 (110, 200), (100, 260), (90, 320)])
def test_prune_odd_shapes_and_all_screen_variants(fc, n, a):
    """Every size class of the screens behind prune_by_rmsd: the split-half kernel with 1 ... 6 k-steps of 32 atoms (up to
    192 atoms; 4 and more: one workgroup per CU), the fp32 matrix pipe beyond (193), the fp64 kernels behind them
    (mfma<4> up to 52 atoms, mfma<8> up to 104, the vector screen without an LDS tile above: 200 atoms still have the
    fp32 matrix pipe in front, 260 and 320 -- the poses of two or three docked molecules, firecode/embedder.py:1472-1474 --
    go to the fp64 vector screen: the size class tools/shape_sweep.py reaches at 260 atoms); tiny atom counts (K padded
    to 4) and tiny N"""
    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=90 + a, cluster_size=3)
    S0, R0, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask, ref)
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, grey = ens.simbits(0.5, 1.0)
    from firecode_amd._lib import unpack_bits

    assert np.array_equal(unpack_bits(bits, n), np.triu(S0, 1))


@pytest.mark.parametrize("n,a", [(150, 193), (140, 224), (130, 260), (120, 320), (100, 384), (100, 416)])
def test_prune_large_compact_structures(fc, n, a):
    """Structures of 193 ... 384 atoms with the radius of gyration of folded molecules / docked poses (globules: 4.5-5.5 A;
    the self-avoiding walks of the other tests have 15-20 A and a band too wide for a single-precision screen): the
    split-half screen with its 32-column tile takes them (7 ... 12 k-steps of 32 atoms; 416 atoms: beyond its LDS tile).
    Masks and similarity bits against the oracle."""
    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=700 + a, cluster_size=3, compact=True)
    S0, R0, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    assert 0 < ref.sum() < n
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask, ref)
    from firecode_amd import _lib

    assert _lib.screen_last_kind() == (16 if a <= 384 else 1)
    with fc.DeviceEnsemble(X, center=True) as ens:
        bits, grey = ens.simbits(0.5, 1.0)
    assert np.array_equal(_lib.unpack_bits(bits, n), np.triu(S0, 1))


@pytest.mark.parametrize("n,a,seed", [(140, 214, 2), (130, 224, 2), (120, 260, 2), (110, 288, 2), (100, 329, 4), (90, 360, 2)])
def test_prune_large_extended_structures_on_the_fp32_pipe(fc, n, a, seed):
    """Extended structures (the generator's self-avoiding walks: radius of gyration 15-20 A) of 214 ... 360 atoms: the band
    of the split-half bound is too wide for them and the fp32 kernel's 64-column tile does not fit the LDS -- the fp32
    matrix-pipe screen with a 32-column tile (odd numbers of 12-row runs included: 329 atoms = 83 k-steps, 249 runs).
    Selected here whatever the band (fc_screen_select(32)); masks against the oracle, and the default selection must
    give the same mask whichever screen it takes."""
    from firecode_amd import _lib

    X, atoms, asg = syn.synthetic_ensemble(n, a, seed=seed, cluster_size=3)
    S0, R0, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    assert 0 < ref.sum() < n
    try:
        _lib.screen_select(32)
        _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
        kind = _lib.screen_last_kind()
    finally:
        _lib.screen_select(0)
    assert kind == 32
    assert np.array_equal(mask, ref)
    _, mask2 = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask2, ref)
    # the pairs the screen lets through hold every similar pair (it is a filter): the exact refine's count of similar pairs
    with fc.DeviceEnsemble(X, center=True) as ens:
        try:
            _lib.screen_select(32)
            tk, ts, m3, st = ens.bench_prune(0.5, 1.0, reps=2, want_mask=True)
        finally:
            _lib.screen_select(0)
    assert np.array_equal(m3.astype(bool), ref) and int(st[2]) == int(np.triu(S0, 1).sum())


@pytest.mark.parametrize("cfg", ["valu8x4", "valu4x8"])
def test_valu_screen_kernels_still_agree(fc, cfg, monkeypatch):
    monkeypatch.setenv("FC_SCREEN_CFG", cfg)
    X, atoms, _ = syn.synthetic_ensemble(500, 33, seed=97)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S0))


@pytest.mark.parametrize("seed,nr1,nr2,scale", [(66, 2, 1, 1.8), (67, 1, 2, 1.25), (68, 2, 2, 1.0)])
def test_embed_grid_dedupe_vs_oracle(fc, seed, nr1, nr2, scale):
    """clash test + the sequential in-group rmsd_similarity filter (embeds.py:713-727)"""
    m1, r1, pv1, m2, r2, pv2 = _embed_case(seed, n1=2, n2=3, A1=9, A2=8, nr1=nr1, nr2=nr2)
    # spread the molecules so that a good share of the poses is clash free
    pv1 = pv1 * scale
    pv2 = pv2 * scale
    steps, rr = 5, 45.0
    angles = np.arange(steps + 1) * 2 * rr / steps - rr
    acc, ok = fc.embeds.embed_grid_poses(m1, r1, pv1, m2, r2, pv2, angles, thresh=(1.2 if scale > 1.5 else 3.0), rmsd_thr=1.0)
    conf_ids = o.cartesian_product(range(len(m1)), range(len(m2)))
    ang_ids = o.cartesian_product(range(steps + 1), range(steps + 1))
    ref_acc = np.zeros_like(acc)
    ref_ok = np.zeros_like(ok)
    for c1, c2 in conf_ids:
        for ori in (0, 1):
            angular_poses = []
            for i1, i2 in ang_ids:
                Ra, ta, Rb, tb = o.bimol_pose_transforms(m1[c1], m2[c2], r1, r2, pv1[c1], pv2[c2],
                                                         (angles[i1], angles[i2]), ori)
                pose = o.get_embed([m1[c1], m2[c2]], [Ra, Rb], [ta, tb])
                if o.compenetration_check(pose, ids=[m1.shape[1], m2.shape[1]], thresh=(1.2 if scale > 1.5 else 3.0)):
                    ref_ok[c2, c1, ori, i2, i1] = True
                    if not o.rmsd_similarity(pose, np.array(angular_poses), rmsd_thr=1):
                        angular_poses.append(pose)
                        ref_acc[c2, c1, ori, i2, i1] = True
    assert np.array_equal(ok, ref_ok)
    assert np.array_equal(acc, ref_acc)
    assert 0 < acc.sum() < ok.sum() <= ok.size
    if scale < 1.5:
        assert ok.sum() < ok.size  # some poses clash


@pytest.mark.parametrize("seed,thresh,delta,pairing", [(21, 1.2, 10.0, False), (22, 1.0, 0.35, False), (23, 1.1, 10.0, True)])
def test_bimolecular_cyclical_embed_driver_vs_oracle(fc, seed, thresh, delta, pairing):
    """_fast_bimol_rigid_cyclical_embed (embeds.py:588-750) end to end: several pivots per
    conformer, the pivot-norm skip, the pairings filter, poses and constrained indices in order"""
    from oracle import cyclical_ref as cy

    mols = syn.synthetic_trimolecular(n_conf=(2, 3, 1), n_atoms=(9, 11, 8), seed=seed, pivots_per_conf=(2, 2, 1))[:2]
    angles = o.cartesian_product(range(6), range(6)) * 2 * 45 / 5 - 45
    table = None
    if pairing:
        cum = [list(m["reactive_cumnums"].values()) for m in mols]
        table = {"a": (cum[0][0], cum[1][1])}  # realised by orientation 1 only
    trace = []
    ref_poses, ref_ci = cy.cyclical_embed_bimolecular(_tri_objects(mols), angles, pairings_table=table,
                                                      clash_thresh=thresh, max_norm_delta=delta, trace=trace)
    poses, ci = fc.embeds.cyclical_embed_bimolecular(mols, angles, pairings_table=table, clash_thresh=thresh,
                                                     max_norm_delta=delta)
    assert poses.shape == ref_poses.shape and len(poses) > 0
    assert np.abs(poses - ref_poses).max() < TOL
    assert np.array_equal(ci, ref_ci)
    n_jobs_all = 2 * 3 * 2 * 2
    if delta < 1.0:
        assert 0 < len({(t[0], t[1]) for t in trace}) < n_jobs_all  # some pivot pairs were skipped
    if pairing:
        assert {t[2] for t in trace} == {1}
    with pytest.raises(fc.FirecodeHipInputError):
        fc.embeds.cyclical_embed_bimolecular(mols, angles[::-1], clash_thresh=thresh)


# ---------------------------------------------------------------- a14: string embed
@pytest.mark.parametrize("seed,thresh", [(75, 1.2), (76, 2.2)])
def test_string_embed_vs_oracle(fc, seed, thresh):
    rng = np.random.default_rng(seed)
    m1 = rng.normal(scale=1.5, size=(3, 9, 3))
    m2 = rng.normal(scale=1.5, size=(2, 7, 3))
    r1, r2 = 2, 4
    c1 = m1[:, [r1]] * 1.7 + rng.normal(scale=0.2, size=(3, 1, 3))
    v1 = c1 - m1[:, [r1]]
    c2 = np.concatenate([m2[:, [r2]] * 1.7, m2[:, [r2]] * -0.9 + 0.3], axis=1)
    v2 = c2 - m2[:, [r2]]
    v2[0, 1] = -v1[0, 0] * 2.0     # already anti-parallel to ref_vec of conformer 0: the identity branch
    v2[1, 0] = v1[1, 0] * 0.5      # exactly parallel to ref_vec: the 180-degree-about-z branch
    angles = np.arange(12) * 30.0
    quads = np.array([[0, 1, r1, 9 + r2], [1, r1, 9 + r2, 9 + 5], [r1, 9 + r2, 9 + 5, 9 + 6]])
    poses, acc, ok = fc.embeds.string_embed_poses(m1, c1, v1, m2, c2, v2, angles, quads, thresh=thresh)
    ok0, acc0, poses0 = o.string_embed(m1, m2, c1, v1, c2, v2, angles, quads, thresh=thresh)
    assert np.array_equal(ok, ok0)
    assert np.array_equal(acc, acc0)
    assert poses.shape == poses0.shape and np.abs(poses - poses0).max() < TOL
    assert 0 < acc.sum() <= ok.sum() <= len(ok)
    if thresh > 2:
        assert ok.sum() < len(ok) and acc.sum() < ok.sum()


def test_gpu_prune_operator(fc, tmp_path, monkeypatch):
    """operator contract: f(filename, embedder) -> output .xyz name"""
    from types import SimpleNamespace

    from firecode_amd import operators as ops

    monkeypatch.chdir(tmp_path)
    X, atoms, asg = syn.synthetic_ensemble(300, 14, seed=77)
    logs = []
    emb = SimpleNamespace(mols={"mol.xyz": SimpleNamespace(coords=X, atoms=atoms, basename="mol")},
                          options=SimpleNamespace(rmsd=0.5, dryrun=False), log=logs.append, debuglog=None)
    out = ops.operate("mol.xyz", "gpu_prune", emb)
    assert out == "mol_gpu_pruned.xyz"
    a, c = fc._lib.xyz_read(tmp_path / out)
    _, m1 = o.prune_by_moment_of_inertia(X, atoms)
    _, m2 = o.prune_by_rmsd(X[m1], atoms, 0.5)
    assert len(c) == m2.sum() and np.abs(c - np.round(X[m1][m2], 6)).max() < 1e-6
    assert any("Discarded" in s for s in logs)
    emb.options.dryrun = True
    assert ops.operate("mol.xyz", "gpu_prune", emb) == "mol.xyz"


def test_gpu_prune_operator_runs_the_third_stage_with_the_molecule_graph(fc, tmp_path, monkeypatch):
    """firecode/operators.py:613-632: MOI -> RMSD -> prune_by_rmsd_rot_corr(..., mol.graph, ...) below 1000
    structures.  A chain ending in a tBu-like C(CH3)3 group whose conformers differ by 120-degree turns of that
    group: the plain RMSD stage keeps them apart, the symmetry-corrected stage (given the graph) merges them."""
    from types import SimpleNamespace

    from firecode_amd import operators as ops
    from firecode_amd import torsion_perception as tp

    monkeypatch.chdir(tmp_path)
    # C0-C1-C2-C3(-C4)(-C5)(-C6): tetrahedral quaternary carbon C3, three methyl carbons (heavy atoms only)
    t = 1.54
    c3 = np.array([0.0, 0.0, 0.0])
    up = np.array([0.0, 0.0, 1.0])
    ring = [np.array([np.cos(a), np.sin(a), 0.0]) for a in (0.3, 0.3 + 2 * np.pi / 3, 0.3 + 4 * np.pi / 3)]
    methyls = [c3 + t * (0.3333 * up + 0.9428 * r) for r in ring]
    base = np.array([c3 - 3 * t * up + np.array([1.2, 0.4, 0]), c3 - 2 * t * up + np.array([0.5, -0.3, 0]), c3 - t * up, c3] + methyls)
    atoms = np.array(["C"] * 7)
    rng = np.random.default_rng(5)
    X = []
    for k in range(12):
        x = base + rng.normal(scale=0.01, size=base.shape)
        # distort ONE methyl so that a plain 120-degree turn is not a relabelling the RMSD stage could undo
        x[4] += np.array([0.0, 0.0, 0.35])
        ang = (k % 3) * 2 * np.pi / 3
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        x[4:] = (x[4:] - c3) @ R.T + c3   # the tBu end turned about the C2-C3 axis
        X.append(x @ _rot(rng).T + rng.normal(scale=3.0, size=3))
    X = np.array(X)
    g = tp.graphize(atoms, X[0])
    assert g.number_of_edges() == 6 and any(t_[1:3] in ((2, 3), (3, 2)) for t_ in tp.symmetric_torsions(g))
    logs = []
    mol = SimpleNamespace(coords=X, atoms=atoms, basename="tbu", graph=g)
    emb = SimpleNamespace(mols={"tbu.xyz": mol}, options=SimpleNamespace(rmsd=0.25, dryrun=False), log=logs.append, debuglog=logs.append)
    out = ops.operate("tbu.xyz", "gpu_prune", emb)
    _, c = fc._lib.xyz_read(tmp_path / out)
    # oracle: the same three stages
    _, m1 = o.prune_by_moment_of_inertia(X, atoms)
    _, m2 = o.prune_by_rmsd(X[m1], atoms, 0.25)
    tors = tp.symmetric_torsions(g, X[0], atoms)
    quads = [t_[:4] for t_ in tors]
    masks = [fc.pruner.rotation_mask(g, q, len(atoms)) for q in quads]  # == the reference's _get_rotation_mask (golden rotmask_out)
    sets = [{2: (0, 180), 3: (0, 120, 240), 4: (0, 90, 180, 270), 6: (0, 60, 120, 180, 240, 300)}[t_[4]] for t_ in tors]
    _, m3 = o.prune_by_rmsd_rot_corr(X[m1][m2], atoms, quads, masks, sets, max_rmsd=0.25)
    assert len(c) == m3.sum() < m2.sum()            # the third stage removed structures the second kept
    assert np.abs(c - np.round(X[m1][m2][m3], 6)).max() < 1e-6
    mol.graph = None                                 # no graph: the first two stages only
    out = ops.operate("tbu.xyz", "gpu_prune", emb)
    assert len(fc._lib.xyz_read(tmp_path / out)[1]) == m2.sum()


def test_rmsd_values_matrix(fc):
    """all-pairs RMSD values from the Newton / MFMA kernel (+ exact fix-up of tiny rmsd)"""
    X, atoms, asg = syn.synthetic_ensemble(700, 50, seed=81)
    X[5] = X[4] @ syn.random_rotation(np.random.default_rng(1)).T + 2.0        # identical up to a rigid motion
    X[9] = X[8] + 1e-6 * np.random.default_rng(2).normal(size=X[8].shape)      # almost identical
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, ms = ens.rmsd_values()
    _, R0, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    assert np.abs(R - R0).max() < TOL
    assert R[4, 5] < 1e-12 and abs(R[8, 9] - R0[8, 9]) < 1e-13 and ms > 0
    # random, unrelated structures (no cluster structure at all), odd sizes
    rng = np.random.default_rng(3)
    Y = rng.normal(scale=2.5, size=(150, 23, 3))
    with fc.DeviceEnsemble(Y, center=True) as ens:
        R, _ = ens.rmsd_values()
    iu, ju = np.triu_indices(150, 1)
    r0, _ = o.rmsd_and_max_batch(Y[iu], Y[ju], center=True)
    assert np.abs(R[iu, ju] - r0).max() < TOL


def test_rmsd_and_max_all_pairs_tiled_kernel(fc):
    """fc_ensemble_rmsd_and_max_all (covariance on the fp64 matrix pipe, rotation + explicit
    difference in the epilogue) against the oracle, also where the rotation is not unique:
    duplicated conformers, a planar and a collinear structure, mirror images"""
    rng = np.random.default_rng(77)
    X, atoms, _ = syn.synthetic_ensemble(157, 23, seed=41)
    X[5] = X[4]                                   # identical pair: rmsd 0
    X[9] = X[8] @ _rot(rng).T + 3.0               # same structure, moved
    X[20, :, 2] = 0.0                             # planar
    X[21] = np.outer(np.linspace(-8, 8, 23), [1.0, 0.5, -0.2])  # collinear
    X[30] = X[31] * np.array([1.0, 1.0, -1.0])    # mirror image of its neighbour
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D, ms = ens.rmsd_and_max_all()
        R2, D2 = ens.rmsd_matrix()
        assert ms > 0 and ens.rmsd_and_max_all(want_matrices=False)[0] is None
    iu, ju = np.triu_indices(len(X), 1)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(R[iu, ju] - r0).max() < TOL and np.array_equal(R, R2) and np.array_equal(D, D2)
    assert np.allclose(R, R.T) and np.all(np.diag(R) == 0) and np.allclose(D, D.T)
    # the max deviation is only defined up to the choice among equally good rotations where the optimum is
    # degenerate (the collinear structure: bound = inf); everywhere else 1e-10 + the pair's conditioning bound
    bound = o.rotation_error_bound_batch(X[iu], X[ju], center=True)
    assert np.all(np.isinf(bound) == ((iu == 21) | (ju == 21)))
    assert np.all(np.abs(D[iu, ju] - d0) <= TOL + bound)
    assert (bound < TOL).mean() > 0.95
    assert R[4, 5] < 1e-12 and R[8, 9] < 1e-7


@pytest.mark.parametrize("n,a", [(140, 104), (130, 105), (120, 160), (100, 260), (90, 320)])
def test_rmsd_and_max_all_pairs_above_the_tiled_kernel(fc, n, a):
    """the complete alignments of all pairs on both sides of the tiled kernel's last atom count (104: the 64-column
    tile plus the kernel's own arrays fill the LDS) and at the sizes of docked poses -- k_matrix_exact takes over, same
    1e-10 against the oracle"""
    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=500 + a, cluster_size=3)
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D, _ = ens.rmsd_and_max_all()
    iu, ju = np.triu_indices(n, 1)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    bound = o.rotation_error_bound_batch(X[iu], X[ju], center=True)
    assert np.abs(R[iu, ju] - r0).max() < TOL
    assert np.all(np.abs(D[iu, ju] - d0) <= TOL + bound) and (bound < TOL).mean() > 0.95
    assert np.allclose(R, R.T) and np.all(np.diag(R) == 0)


@pytest.mark.parametrize("n,a,kind", [(700, 50, "clusters"), (600, 23, "continuous"), (500, 64, "continuous"), (400, 80, "clusters"),
                                      (300, 130, "clusters"), (200, 260, "clusters"), (500, 40, "duplicates"), (400, 24, "mirror"),
                                      (450, 30, "far"), (420, 32, "large")])
def test_complete_alignments_rmsd_from_the_eigenvalue_equals_the_running_sum(fc, monkeypatch, n, a, kind):
    """The tiled kernel's rmsd comes from the largest eigenvalue, (Gp + Gq) - 2 lambda (k_simbits_screen_mfma<.., EIG>);
    FC_COMPLETE_EIG=0 keeps the atom pass's running sum.  Both forms on clusters, continuous ensembles, exact and 1e-5 A
    duplicates (closer than ~1e-3 A: the eigenvalue form hands the pair to the fix-up kernel), a mirror-symmetric
    structure, coordinates far from the origin and 40 x larger, at every column-tile width (64 / 32 / 16): rmsd within 1e-11
    of each other (2e-11 x the scale for the large coordinates) and within 1e-10 of the oracle, the maximum deviation --
    the explicit rotated difference in both -- bit for bit equal except where the fix-up kernel took the pair."""
    rng = np.random.default_rng(1000 + n + a)
    if kind == "clusters":
        X = syn.synthetic_ensemble(n, a, seed=n + a, cluster_size=4)[0]
    elif kind in ("continuous", "far", "large"):
        X = syn.continuous_ensemble(n, a, seed=n + a)
        if kind == "far":
            X = X + np.array([250.0, -90.0, 40.0])
        if kind == "large":
            X = X * 40.0
    elif kind == "duplicates":
        X = syn.continuous_ensemble(n, a, seed=n + a)
        X[200:350] = X[:150]
        X[350:] = X[:150] + rng.normal(scale=1e-5, size=(150, a, 3))
    else:  # atoms that tie for the largest deviation
        base = rng.normal(scale=2.0, size=(a // 2, 3))
        X = np.repeat(np.concatenate([base, base * np.array([1.0, 1.0, -1.0])])[None], n, axis=0).copy()
        amp = rng.normal(scale=0.3, size=(n, a // 2, 3))
        X[:, : a // 2] += amp
        X[:, a // 2:] += amp * np.array([1.0, 1.0, -1.0])
    scale = 40.0 if kind == "large" else 1.0
    with fc.DeviceEnsemble(X, center=True) as ens:
        R1, D1, _ = ens.rmsd_and_max_all()
        _, _, st1 = ens.bench_rmsd_and_max_all(1)
        monkeypatch.setenv("FC_COMPLETE_EIG", "0")
        R0, D0, _ = ens.rmsd_and_max_all()
        _, _, st0 = ens.bench_rmsd_and_max_all(1)
        monkeypatch.delenv("FC_COMPLETE_EIG")
    iu, ju = np.triu_indices(n, 1)
    assert np.abs(R1 - R0)[iu, ju].max() < 1e-11 * scale * scale
    same = D1[iu, ju] == D0[iu, ju]
    assert np.abs(D1 - D0)[iu, ju].max() < 1e-11 * scale and (~same).sum() <= int(st1[1]) + int(st0[1])
    assert int(st1[1]) >= int(st0[1])
    if kind == "duplicates":
        assert int(st1[1]) >= 150 + 150 and R1[0, 200] < 1e-12 and 0 < R1[0, 350] < 1e-4
    else:
        assert int(st1[1]) <= 5
    sel = rng.choice(len(iu), size=3000, replace=False)
    r0, d0 = o.rmsd_and_max_batch(X[iu[sel]], X[ju[sel]], center=True)
    bound = o.rotation_error_bound_batch(X[iu[sel]], X[ju[sel]], center=True)
    assert np.abs(R1[iu[sel], ju[sel]] - r0).max() < TOL * scale
    assert np.all(np.abs(D1[iu[sel], ju[sel]] - d0) <= TOL * scale + bound)


@pytest.mark.parametrize("n,a", [(2, 1), (5, 1), (70, 2), (130, 3), (65, 4), (200, 5), (3, 50), (129, 7)])
def test_complete_alignments_of_tiny_structures(fc, n, a):
    """one to seven atoms, two conformers: rank-deficient covariances (every pair of a one- or two-atom structure goes to the
    fix-up kernel), a single row tile, a single pair -- the tiled kernel's outputs against the oracle, no NaN anywhere"""
    rng = np.random.default_rng(3 * n + a)
    X = rng.normal(scale=2.0, size=(n, a, 3))
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D, _ = ens.rmsd_and_max_all()
    iu, ju = np.triu_indices(n, 1)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    bound = o.rotation_error_bound_batch(X[iu], X[ju], center=True)
    assert not np.isnan(R).any() and not np.isnan(D).any()
    assert np.abs(R[iu, ju] - r0).max() < TOL
    fin = np.isfinite(bound)
    assert np.all(np.abs(D[iu, ju] - d0)[fin] <= TOL + bound[fin])
    assert np.allclose(R, R.T) and np.all(np.diag(R) == 0)


def test_complete_alignments_random_shapes_both_forms(fc, monkeypatch):
    """A seeded sweep over shapes the directed cases do not name one by one: 2 - 700 conformers, 1 - 300 atoms (all three tile
    widths and the exact kernel's side of 416), clustered / continuous / unrelated / duplicated / planar / mirrored structures
    with random rigid motions -- the eigenvalue form against the running sum (1e-11), both against the oracle on a sample."""
    rng = np.random.default_rng(20261005)
    for case in range(30):
        n = int(rng.choice([2, 3, 17, 64, 65, 129, 200, 257, 400, 700]))
        a = int(rng.choice([1, 2, 3, 4, 7, 16, 33, 50, 52, 53, 80, 104, 105, 140, 208, 209, 300]))
        if n * n * a > 6e7:
            n = 129
        kind = str(rng.choice(["clusters", "continuous", "random", "duplicates", "planar", "mirror"]))
        seed = int(rng.integers(1 << 30))
        if kind == "clusters" and a >= 3:
            X = syn.synthetic_ensemble(max(n, 5), a, seed=seed, cluster_size=3 if a > 100 else 5)[0][:n]
        elif kind == "continuous" and a >= 3:
            X = syn.continuous_ensemble(n, a, seed=seed)
        else:
            X = rng.normal(scale=1.5, size=(n, a, 3))
            if kind == "duplicates":
                X[n // 2:] = X[: n - n // 2] + rng.normal(scale=10.0 ** rng.uniform(-7, -2), size=(n - n // 2, a, 3))
            elif kind == "planar":
                X[:, :, 2] = 0.0
            elif kind == "mirror":
                X[1::2] = X[0::2][: len(X[1::2])] * np.array([1.0, 1.0, -1.0])
        X = np.einsum("nij,naj->nai", np.array([_rot(rng) for _ in range(n)]), X) + rng.normal(scale=3.0, size=(n, 1, 3))
        with fc.DeviceEnsemble(X, center=True) as ens:
            R1, D1, _ = ens.rmsd_and_max_all()
            monkeypatch.setenv("FC_COMPLETE_EIG", "0")
            R0, D0, _ = ens.rmsd_and_max_all()
            monkeypatch.delenv("FC_COMPLETE_EIG")
        iu, ju = np.triu_indices(n, 1)
        where = (case, n, a, kind)
        assert not np.isnan(R1).any() and not np.isnan(D1).any(), where
        assert np.abs(R1 - R0)[iu, ju].max() < 1e-11, where
        sel = rng.choice(len(iu), size=min(len(iu), 1500), replace=False)
        r0, d0 = o.rmsd_and_max_batch(X[iu[sel]], X[ju[sel]], center=True)
        bound = o.rotation_error_bound_batch(X[iu[sel]], X[ju[sel]], center=True)
        assert np.abs(R1[iu[sel], ju[sel]] - r0).max() < TOL, where
        fin = np.isfinite(bound)
        assert np.all(np.abs(D1[iu[sel], ju[sel]] - d0)[fin] <= TOL + bound[fin]), where
        assert np.all(np.abs(D0[iu[sel], ju[sel]] - d0)[fin] <= TOL + bound[fin]), where


def test_complete_alignments_of_near_duplicates_beyond_the_fix_up_queue(fc, monkeypatch):
    """An ensemble of copies: every pair is closer than the eigenvalue form of the rmsd can resolve and the fix-up queue
    (here cut to 1 000 entries) overflows -- fc_ensemble_rmsd_and_max_all then runs the tiled kernel again with the
    running sum, which declines no pair of these."""
    rng = np.random.default_rng(5)
    base = syn.continuous_ensemble(1, 40, seed=3)[0]
    X = base[None] + rng.normal(scale=2e-6, size=(300, 40, 3))
    monkeypatch.setenv("FC_PAIRQ_CAP", "1000")
    with fc.DeviceEnsemble(X, center=True) as ens:
        R, D, _ = ens.rmsd_and_max_all()
    iu, ju = np.triu_indices(300, 1)
    r0, d0 = o.rmsd_and_max_batch(X[iu], X[ju], center=True)
    assert np.abs(R[iu, ju] - r0).max() < TOL and np.abs(D[iu, ju] - d0).max() < TOL
    assert r0.max() < 1e-4 and np.allclose(R, R.T) and np.all(np.diag(R) == 0)


def test_context_lifecycle_and_threads(fc):
    """fc_shutdown / fc_init: streams, events and staging are rebuilt, ensembles of the old context
    are refused (never used), pipelined prunes work again; two host threads may call the library
    (their calls are serialised by its lock)"""
    import threading

    from firecode_amd import _lib

    X, atoms, asg = syn.synthetic_ensemble(600, 20, seed=303)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    old = fc.DeviceEnsemble(X, center=True)
    old.bench_prune(0.5, 1.0, reps=4)            # side streams and the event pool exist now

    def ladder(n=140000):                        # ... and the TFD ladder's helper streams and pinned staging pieces
        rng = np.random.default_rng(9)
        i = np.arange(n, dtype=np.int64)
        fm = np.where(rng.random(n) < 0.9, i + rng.integers(1, 60, n), -1)
        fm[fm >= n] = -1
        out = {}
        for env in ("1", "0"):
            os.environ["FC_TFD_GPU"] = env
            m = np.zeros(n, dtype=np.uint8)
            _lib.call("fc_tfd_ladder_from_first_match", _lib.pi(fm), n, _lib.pb(m))
            out[env] = m
        os.environ.pop("FC_TFD_GPU")
        assert np.array_equal(out["0"], out["1"]) and 0 < out["0"].sum() < n
        return out["1"]

    before = ladder()
    _lib.shutdown()
    _lib.init(0)
    assert np.array_equal(ladder(), before)
    with pytest.raises(_lib.FirecodeHipInputError):
        old.prune(0.5, 1.0)
    old.close()
    with fc.DeviceEnsemble(X, center=True) as ens:
        _, _, mask, _ = ens.bench_prune(0.5, 1.0, reps=4)
        assert np.array_equal(mask, ref)
    results, errors = {}, []

    def worker(k):
        try:
            with fc.DeviceEnsemble(X, center=True) as e:
                for _ in range(5):
                    m, _ = e.prune(0.5, 1.0)
                    _, _, m2, _ = e.bench_prune(0.5, 1.0, reps=3)
                    assert np.array_equal(m, ref) and np.array_equal(m2, ref)
            results[k] = True
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors and len(results) == 3


def test_convention_switches_reach_the_kernels(fc):
    """firecode_amd.pruner.CONVENTIONS: each switch against the oracle's same-named switch --
    '<=' as '<' against the next double, the mirror drop rule in the pair ladder, the factor,
    the window and the default threshold"""
    from firecode_amd import pruner

    X, atoms, _ = syn.synthetic_ensemble(400, 14, seed=21)
    en = np.random.default_rng(5).uniform(0, 2, len(X))
    Xz = X - X.mean(axis=1, keepdims=True)
    r01 = o.rmsd_and_max(Xz[0], Xz[1])[0]
    saved = dict(pruner.CONVENTIONS)
    try:
        for kw in (dict(drop="later"), dict(strict_lt=False), dict(maxdev_factor=1.0), dict(window_strict=False),
                   dict(drop="later", strict_lt=False, maxdev_factor=3.0)):
            pruner.CONVENTIONS.update(saved)
            pruner.CONVENTIONS.update(kw)
            okw = {k: v for k, v in kw.items()}
            assert np.array_equal(pruner.prune_by_rmsd(X, atoms, 0.5)[1], o.prune_by_rmsd(X, atoms, 0.5, **okw)[1]), kw
            assert np.array_equal(pruner.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0)[1],
                                  o.prune_by_rmsd(X, atoms, 0.5, energies=en, max_dE=1.0, **okw)[1]), kw
        # "<=" reaches the kernels as "<" against the next double up (a tie itself cannot be staged across
        # two implementations whose rmsd agree to 1e-15, not to the last bit)
        pruner.CONVENTIONS.update(saved)
        pruner.CONVENTIONS["strict_lt"] = False
        assert pruner._thresholds(r01, 10.0, 1.0) == (np.nextafter(r01, np.inf), np.nextafter(10.0, np.inf), 1.0)
        pruner.CONVENTIONS.update(strict_lt=True, window_strict=False)
        assert pruner._thresholds(r01, None, 1.0) == (r01, 2.0 * r01, np.nextafter(1.0, np.inf))
        pruner.CONVENTIONS.update(saved)
        pruner.CONVENTIONS["default_max_rmsd"] = 0.4
        assert np.array_equal(pruner.prune_by_rmsd(X, atoms)[1], o.prune_by_rmsd(X, atoms, 0.4)[1])
    finally:
        pruner.CONVENTIONS.clear()
        pruner.CONVENTIONS.update(saved)
        pruner._thresholds(0.5, None, 0.0)  # back to the default drop rule in the library


def test_fused_similarity_pipeline_equals_stage_by_stage(fc):
    """fc_prune_similarity (MOI -> gather on the device -> RMSD on ONE upload) against the two
    stand-alone functions and against the oracle's composition: with and without energies (ties
    included: stable order), hydrogens excluded from the RMSD stage only, every stage mask"""
    from firecode_amd import pruner

    rng = np.random.default_rng(17)
    X, _, asg = syn.synthetic_ensemble(700, 16, seed=301)
    X[100:140] = X[100:140] * 1.05  # same shape, scaled: MOI tells these apart from their cluster mates
    atoms = np.array(["C", "H", "N", "C", "O", "H", "C", "C"] * 2)
    en = np.round(rng.uniform(0, 2, len(X)), 1)  # many exact ties
    for energies, dE in ((None, 0.0), (en, 1.0), (en, 0.3)):
        kw = {} if energies is None else dict(energies=energies, max_dE=dE)
        m_moi, m_both, counts = pruner.prune_similarity(X, atoms, max_rmsd=0.5, **kw)
        s1, a = pruner.prune_by_moment_of_inertia(X, atoms, **kw)
        kw2 = {} if energies is None else dict(energies=energies[a], max_dE=dE)
        _, b = pruner.prune_by_rmsd(s1, atoms, 0.5, **kw2)
        ref = np.zeros(len(X), dtype=bool)
        ref[np.flatnonzero(a)[b]] = True
        assert np.array_equal(m_moi, a) and np.array_equal(m_both, ref)
        assert counts.tolist() == [len(X), int(a.sum()), int(ref.sum())]
        _, oa = o.prune_by_moment_of_inertia(X, atoms, **kw)
        _, ob = o.prune_by_rmsd(X[oa], atoms, 0.5, **kw2)
        oref = np.zeros(len(X), dtype=bool)
        oref[np.flatnonzero(oa)[ob]] = True
        assert np.array_equal(m_moi, oa) and np.array_equal(m_both, oref)
    # single stages through the same entry point
    assert np.array_equal(pruner.prune_similarity(X, atoms, moi=True, rmsd=False)[1], pruner.prune_by_moment_of_inertia(X, atoms)[1])
    assert np.array_equal(pruner.prune_similarity(X, atoms, moi=False, rmsd=True, max_rmsd=0.5)[1],
                          pruner.prune_by_rmsd(X, atoms, 0.5)[1])
    # the drivers go through it (same masks as before) and still log one line per stage
    lines = []
    ens = fc.ensemble.Ensemble(atoms=atoms, coords=X.copy(), energies=en.copy(), basename="fused", logfunction=lines.append)
    ens.similarity_pruning(max_rmsd=0.5)
    _, m_b, _ = pruner.prune_similarity(X, atoms, max_rmsd=0.5, energies=en, max_dE=1.0)
    assert np.array_equal(ens.coords, X[m_b]) and np.array_equal(ens.energies, en[m_b])
    assert any("MOI similarity" in ln for ln in lines) and any("RMSD similarity" in ln for ln in lines)


@pytest.mark.parametrize("case", ["grid", "subset_dups", "negative"])
def test_torsion_scan_prefix_tree_equals_row_kernel(fc, monkeypatch, case):
    """The scan as a prefix tree over the sorted angle-sets (default for >= 4096 rows that share
    prefixes) against the one-wavefront-per-row kernel (FC_SCAN_TREE=0): identical bits for
    conformers, rotated-bond counts and fingerprints -- full grid in cartesian_product order, a random
    subset with duplicate rows, negative and back-off-prone angles; and the oracle on a sample"""
    base, tors, masks = _chain_case(34, 5, seed=77)
    rng = np.random.default_rng(5)
    if case == "grid":
        angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 5)          # 7776 rows
    elif case == "subset_dups":
        grid = o.cartesian_product((0, 90, 180, 270), (0, 120, 240), (0, 60, 120, 180, 240, 300), (0, 180), (0, 120, 240))
        angles = np.concatenate([grid, grid[rng.integers(0, len(grid), 5000)]])[rng.permutation(len(grid) + 5000)]
    else:
        angles = o.cartesian_product((-170, -45, 0, 33, 170), (0, 7, 120), (-90, 0, 90), (0, 45, 200, 355), (0, -5, 5, 10))
        angles = np.concatenate([angles] * 6)[:5400]
    quads = tors
    tf, rot, out = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, quads, thresh=1.5, want_coords=True)
    monkeypatch.setenv("FC_SCAN_TREE", "0")
    tf0, rot0, out0 = fc.torsion_module.torsion_scan_fingerprints(base, tors, masks, angles, quads, thresh=1.5, want_coords=True)
    assert np.array_equal(rot, rot0) and np.array_equal(out, out0) and np.array_equal(tf, tf0)
    pick = rng.choice(len(angles), 60, replace=False)
    o_out, o_rot = o.torsion_scan(base, tors, masks, angles[pick], thresh=1.5)
    assert np.array_equal(rot[pick], o_rot) and np.abs(out[pick] - o_out).max() < TOL


# ---------------------------------------------------------------- reference-signature adapters
def test_csearch_adapters_with_the_reference_signatures(fc):
    """clustered_csearch(atoms, coords, torsions, graph, ...) / random_csearch(...) as FIRECODE's csearch calls them
    (firecode/torsion_module.py:697-723) with duck-typed Torsion objects: rotation masks from this package's
    rotation_mask over the bond graph, output == the numeric core == the oracle pipeline; log lines through logfunction"""
    import networkx as nx
    from types import SimpleNamespace

    base, tors, masks = _chain_case(26, 3, seed=73)
    graph = nx.path_graph(26)
    atoms = np.array(["C"] * 26)
    torsions = [SimpleNamespace(torsion=tuple(int(i) for i in t), n_fold=6) for t in tors]
    for t, m in zip(torsions, masks):
        assert np.array_equal(fc.torsion_module._get_rotation_mask(graph, t.torsion), m)
    log = []
    out = fc.torsion_module.clustered_csearch(atoms, base, torsions, graph, constrained_indices=None, n=100, n_out=10_000,
                                              title="chain", logfunction=log.append, interactive_print=False,
                                              write_torsions=False, debug=False)
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * 3)
    sc, rot = o.torsion_scan(base, tors, masks, angles)
    ref, _ = o.prune_conformers_tfd(np.concatenate([base[None], sc[rot != 0]]), tors)
    assert out.shape == ref.shape and np.abs(out - ref).max() < TOL
    assert any("Clustered CSearch on chain" in line for line in log) and any("6-fold" in line for line in log)
    # random search: the reference shuffles with the global generator; `order` fixes the permutation
    perm = np.random.default_rng(5).permutation(len(angles))
    got = fc.torsion_module.random_csearch(atoms, base, torsions, graph, n_out=25, title="chain", logfunction=None,
                                           interactive_print=False, order=perm)
    ref_r, _ = o.random_csearch(base, tors, masks, angles[perm], n_out=25)
    assert got.shape == ref_r.shape and np.abs(got - ref_r).max() < TOL
    with pytest.raises(NotImplementedError):
        fc.torsion_module.clustered_csearch(atoms, base, torsions, graph, write_torsions=True, logfunction=None)
    with pytest.raises(fc.FirecodeHipInputError):
        fc.torsion_module.clustered_csearch(atoms, base, [SimpleNamespace(torsion=(0, 1, 2, 3), n_fold=5)], graph, logfunction=None)


class _FakeEmbedder:
    """what cyclical_embed / string_embed read from FIRECODE's Embedder (firecode/embedder.py), nothing more"""

    def __init__(self, objects, angles, clash_thresh, pairings_table=None):
        from types import SimpleNamespace

        self.objects = objects
        self.systematic_angles = angles
        self.options = SimpleNamespace(clash_thresh=clash_thresh, debug=False)
        self.pairings_table = pairings_table or {}
        self.internal_constraints = []
        self.ids = [len(m.coords[0]) for m in objects]
        self.embed, self.candidates, self.lines = "cyclical", 0, []

    def log(self, msg="", p=True):
        self.lines.append(msg)


def _hypermolecules(mols):
    """Hypermolecule-shaped objects (coords, reactive_indices, pivots of Pivot-shaped objects, reactive_atoms_classes_dict)
    from synthetic_trimolecular's dictionaries"""
    from types import SimpleNamespace

    out = []
    for m in mols:
        atoms_of = {int(c): int(i) for i, c in m["reactive_cumnums"].items()}
        ratom = {int(i): SimpleNamespace(cumnum=int(c)) for i, c in m["reactive_cumnums"].items()}
        pivots = [[SimpleNamespace(start=np.asarray(s), end=np.asarray(e), pivot=np.asarray(s) - np.asarray(e),
                                   meanpoint=(np.asarray(s) + np.asarray(e)) / 2, start_atom=ratom[atoms_of[int(cs)]],
                                   end_atom=ratom[atoms_of[int(ce)]]) for s, e, cs, ce in plist] for plist in m["pivots"]]
        out.append(SimpleNamespace(coords=m["coords"], reactive_indices=m["reactive_indices"], pivots=pivots,
                                   reactive_atoms_classes_dict={0: ratom}))
    return out


@pytest.mark.parametrize("n_mols", [2, 3])
def test_cyclical_embed_adapter_with_the_reference_signature(fc, n_mols):
    """cyclical_embed(embedder, max_norm_delta) (firecode/embeds.py:180): reads the embedder's molecules, angles, pairings
    and clash threshold, sets embedder.constrained_indices, returns the poses of the numeric cores (which are checked
    against the literal oracle elsewhere); raises ZeroCandidatesError when nothing fits"""
    mols = syn.synthetic_trimolecular(n_conf=(2, 1, 2), n_atoms=(9, 12, 8), seed=3, pivots_per_conf=(1, 2, 1))[:n_mols]
    if n_mols == 2:
        angles = o.cartesian_product(range(6), range(6)) * 2 * 45 / 5 - 45
        ref, ref_ci = fc.embeds.cyclical_embed_bimolecular(mols, angles, clash_thresh=0.9, max_norm_delta=5.0)
    else:
        angles = o.cartesian_product(range(3), range(3), range(3)) * 2 * 45 / 2 - 45
        ref, ref_ci = fc.embeds.cyclical_embed_trimolecular(mols, angles, clash_thresh=0.9)
    emb = _FakeEmbedder(_hypermolecules(mols), angles, 0.9)
    poses = fc.embeds.cyclical_embed(emb, max_norm_delta=5.0)
    assert len(poses) > 0 and np.array_equal(poses, ref) and np.array_equal(emb.constrained_indices, ref_ci)
    assert any("embed" in line for line in emb.lines)
    tight = _FakeEmbedder(_hypermolecules(mols), angles, 50.0)  # every pose clashes at 50 A
    with pytest.raises(fc.embeds.ZeroCandidatesError):
        fc.embeds.cyclical_embed(tight)


def test_string_embed_adapter_with_the_reference_signature(fc):
    """string_embed(embedder) (firecode/embeds.py:51): orbitals from mol.get_r_atoms(c)[0], quadruplets of the joined bond
    graph, embedder.constrained_indices repeated per pose (:161-178)"""
    import networkx as nx
    from types import SimpleNamespace

    from firecode_amd.torsion_perception import get_quadruplets

    rng = np.random.default_rng(75)
    skel1, skel2 = syn.synthetic_skeleton(9, rng), syn.synthetic_skeleton(7, rng)
    m1 = skel1[None] + rng.normal(scale=0.05, size=(3, 9, 3))
    m2 = skel2[None] + rng.normal(scale=0.05, size=(2, 7, 3))
    r1, r2 = 2, 4
    c1 = m1[:, [r1]] + 1.2 * (m1[:, [r1]] - m1.mean(axis=1, keepdims=True))
    c2 = m2[:, [r2]] + 1.2 * (m2[:, [r2]] - m2.mean(axis=1, keepdims=True))
    v1, v2 = c1 - m1[:, [r1]], c2 - m2[:, [r2]]

    def mol(coords, centers, vecs, reactive, n):
        g = nx.path_graph(n)
        nx.set_node_attributes(g, {i: "C" for i in range(n)}, "atoms")
        return SimpleNamespace(coords=coords, graph=g, reactive_indices=np.array([reactive]),
                               get_r_atoms=lambda c: [SimpleNamespace(center=centers[c], orb_vecs=vecs[c])])

    angles = np.arange(12) * 30.0
    emb = _FakeEmbedder([mol(m1, c1, v1, r1, 9), mol(m2, c2, v2, r2, 7)], angles, 1.2)
    poses = fc.embeds.string_embed(emb)
    joined = nx.path_graph(9)
    for a, b in nx.path_graph(7).edges():
        joined.add_edge(a + 9, b + 9)
    joined.add_edge(r1, 9 + r2)
    nx.set_node_attributes(joined, {i: "C" for i in range(16)}, "atoms")
    quads = get_quadruplets(joined)
    ref, acc, ok = fc.embeds.string_embed_poses(m1, c1, v1, m2, c2, v2, angles, quads, thresh=1.2)
    ok0, acc0, poses0 = o.string_embed(m1, m2, c1, v1, c2, v2, angles, quads, thresh=1.2)
    assert len(quads) > 0 and len(poses) > 0 and np.array_equal(poses, ref) and np.abs(poses - poses0).max() < TOL
    assert np.array_equal(emb.constrained_indices, np.array([[[r1, 9 + r2]]] * len(poses)))
