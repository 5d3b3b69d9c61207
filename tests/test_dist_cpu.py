"""world_size-2 (and 3) gloo tests of the sharded ladder: the host control
loop, row ownership and the mask exchange of firecode_amd.dist, with the
per-level GPU kernels replaced by an oracle-backed stand-in (CPU only)."""

import os
import socket

import numpy as np
import pytest

from firecode_amd import dist as fdist
from firecode_amd import synthetic as syn
from oracle import cpu_ref as o


class OracleShard:
    """Stands in for DeviceEnsemble.prune_begin / prune_level on one rank:
    same contract (owned rows updated, the others copied from mask_in)."""

    def __init__(self, S, rank, world, row_block):
        self.N = S.shape[0]
        self.S = np.triu(S, 1)
        self.own = fdist.owner_of_rows(self.N, world, row_block) == rank

    def prune_begin(self, *a, **kw):
        return np.zeros(6, dtype=np.int64)

    def prune_level(self, k, mask_in):
        n = self.N
        mask_in = np.asarray(mask_in, dtype=bool)
        out = mask_in.copy()
        chunk = n // k
        for i in np.flatnonzero(self.own & mask_in):
            c = min(i // chunk, k - 1) if chunk > 0 else 0
            last = n if c == k - 1 else chunk * (c + 1)
            if np.any(self.S[i, i + 1:last] & mask_in[i + 1:last]):
                out[i] = False
        return out.astype(np.uint8)


class OraclePairShard(OracleShard):
    """adds the similar-pair exchange (fc_prune_similar_pairs / fc_prune_from_pairs)"""

    def similar_pairs(self, count_hint=None):
        i, j = np.nonzero(self.S & self.own[:, None])
        return (i.astype(np.uint64) << np.uint64(32)) | j.astype(np.uint64)

    def prune_from_pairs(self, pairs, min_per_group=20):
        S = np.zeros((self.N, self.N), dtype=bool)
        ok = pairs != fdist.PAD
        S[(pairs[ok] >> np.uint64(32)).astype(np.int64), (pairs[ok] & np.uint64(0xFFFFFFFF)).astype(np.int64)] = True
        return o.greedy_prune_from_matrix(S | S.T, min_per_group=min_per_group)


def _spawn(fn, args, nprocs, join=True):
    import torch.multiprocessing as mp

    return mp.spawn(fn, args=args, nprocs=nprocs, join=join)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, a, seed, row_block, out_dir, pairs_mode=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist  # lazily: collecting this file must not map torch's HIP runtime into a GPU test session

    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
    S, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    shard = (OraclePairShard if pairs_mode else OracleShard)(S, rank, world, row_block)
    trace = []
    mask, _ = fdist.prune_by_rmsd_sharded(shard, 0.5, rank=rank, world=world,
                                          allgather_fn=fdist.torch_allgather(), row_block=row_block,
                                          trace=trace)
    np.save(os.path.join(out_dir, f"mask_{rank}.npy"), mask)
    np.save(os.path.join(out_dir, f"trace_{rank}.npy"), np.array(trace))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,row_block", [(2, 700, 64), (3, 450, 32), (2, 130, 256)])
def test_sharded_ladder_matches_single(tmp_path, world, n, row_block):
    a, seed = 12, 40 + world
    _spawn(_worker, args=(world, _free_port(), n, a, seed, row_block, str(tmp_path)), nprocs=world, join=True)
    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
    S, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S)
    masks = [np.load(tmp_path / f"mask_{r}.npy") for r in range(world)]
    for m in masks:
        assert np.array_equal(m, ref)
    ref_trace = []
    o.greedy_prune(n, lambda i, j: S[i, j], trace=ref_trace)
    assert np.array_equal(np.load(tmp_path / "trace_0.npy"), np.array(ref_trace))


@pytest.mark.parametrize("world,n", [(2, 600), (3, 333)])
def test_sharded_pairs_exchange_matches_single(tmp_path, world, n):
    a, seed, row_block = 10, 60 + world, 64
    _spawn(_worker, args=(world, _free_port(), n, a, seed, row_block, str(tmp_path), True), nprocs=world, join=True)
    X, atoms, _ = syn.synthetic_ensemble(n, a, seed=seed)
    S, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"mask_{r}.npy"), ref)


def test_gather_pairs_variable_length():
    """3 ranks emulated in-process: one collective when every list fits the capacity,
    a second one when a list is longer; None when a rank has no list"""
    lists = [np.arange(5, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.arange(100, 103, dtype=np.uint64)]

    for cap in (8, 3):
        outs = []
        n_coll = []
        for r in range(3):
            # build the replies exactly as an all-gather would: stack every rank's buffer
            def first(l):
                buf = np.full(cap + 1, fdist.PAD, dtype=np.uint64)
                buf[0] = fdist.NONE if l is None else np.uint64(len(l))
                if l is not None and len(l):
                    buf[1:1 + min(len(l), cap)] = l[:cap]
                return buf.view(np.uint8)
            longest = max(len(l) for l in lists)
            def second(l):
                buf = np.full(longest, fdist.PAD, dtype=np.uint64)
                buf[:len(l)] = l
                return buf.view(np.uint8)
            replies = [np.stack([first(l) for l in lists]), np.stack([second(l) for l in lists])]
            used = []
            def fn(buf):
                used.append(1)
                return replies[len(used) - 1]
            outs.append(fdist.gather_pairs(lists[r], fn, cap))
            n_coll.append(len(used))
        for out in outs:
            assert np.array_equal(out, np.concatenate(lists))
        assert n_coll == ([1, 1, 1] if cap >= 5 else [2, 2, 2])
    # a rank without a list: everybody gets None
    def fn_none(buf):
        rows = []
        for l in (lists[0], None, lists[2]):
            b = np.full(9, fdist.PAD, dtype=np.uint64)
            b[0] = fdist.NONE if l is None else np.uint64(len(l))
            rows.append(b.view(np.uint8))
        return np.stack(rows)
    assert fdist.gather_pairs(lists[0], fn_none, 8) is None


def test_owner_of_rows_balances_triangle():
    for n, rb, tol in ((10000, 128, 1.01), (28284, 256, 1.01), (10000, 256, 1.08)):
        world = 8
        own = fdist.owner_of_rows(n, world, rb)
        work = np.array([(n - 1 - np.flatnonzero(own == r)).sum() for r in range(world)], dtype=float)
        assert work.max() / work.min() < tol
        assert set(np.unique(own)) == set(range(world))


def test_run_ladder_world1_equals_oracle():
    X, atoms, _ = syn.synthetic_ensemble(500, 10, seed=44)
    S, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    shard = OracleShard(S, 0, 1, 64)
    mask = fdist.run_ladder(500, shard.prune_level, lambda m: m[None])
    assert np.array_equal(mask, o.greedy_prune_from_matrix(S))


# ---- the embarrassingly parallel rows: torsion scan and pose grid ---------
def _chain(n_atoms, n_tors, seed):
    rng = np.random.default_rng(seed)
    base = np.cumsum(rng.normal(scale=0.9, size=(n_atoms, 3)) + [1.3, 0, 0], axis=0)
    step = (n_atoms - 4) // n_tors
    tors = np.array([[1 + k * step, 2 + k * step, 3 + k * step, 4 + k * step] for k in range(n_tors)])
    masks = np.zeros((n_tors, n_atoms), dtype=bool)
    for k, t in enumerate(tors):
        masks[k, t[3]:] = True
    return base, tors, masks


def _oracle_grid(m1, reactive1, pivots1, m2, reactive2, pivots2, angles1, angles2=None, thresh=1.5, max_clashes=0):
    """oracle-backed stand-in for embeds.embed_grid_clash (same index order [c2, c1, o, a2, a1])"""
    a1 = np.asarray(angles1, float).reshape(-1)
    a2 = a1 if angles2 is None else np.asarray(angles2, float).reshape(-1)
    out = np.zeros((len(m2), len(m1), 2, len(a2), len(a1)), dtype=bool)
    for c2 in range(len(m2)):
        for c1 in range(len(m1)):
            for orient in (0, 1):
                for i2, ang2 in enumerate(a2):
                    for i1, ang1 in enumerate(a1):
                        R1, t1, R2, t2 = o.bimol_pose_transforms(m1[c1], m2[c2], reactive1, reactive2, pivots1[c1],
                                                                 pivots2[c2], (ang1, ang2), orient)
                        pose = o.get_embed([m1[c1], m2[c2]], [R1, R2], [t1, t2])
                        out[c2, c1, orient, i2, i1] = o.compenetration_check(
                            pose, ids=[m1.shape[1], m2.shape[1]], thresh=thresh, max_clashes=max_clashes)
    return out, 0.0


def _grid_case(seed):
    rng = np.random.default_rng(seed)
    m1 = rng.normal(scale=1.6, size=(3, 7, 3))
    m2 = rng.normal(scale=1.6, size=(5, 6, 3))
    r1, r2 = np.array([0, 3]), np.array([1, 4])
    p1 = np.stack([m1[:, 0] + 0.9, m1[:, 3] - 0.8], axis=1)
    p2 = np.stack([m2[:, 1] + 0.7, m2[:, 4] - 1.0], axis=1)
    return m1, r1, p1, m2, r2, p2, np.array([-45.0, 0.0, 45.0])


def _worker_parallel_rows(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ag = fdist.torch_allgather()
    base, tors, masks = _chain(16, 3, seed=9)
    angles = o.cartesian_product((0, 120, 240), (0, 180), (0, 120, 240))
    out, rot, (lo, hi), keep = fdist.torsion_scan_sharded(base, tors, masks, angles, rank=rank, world=world,
                                                          allgather_fn=ag, scan_fn=o.torsion_scan)
    np.savez(os.path.join(out_dir, f"scan_{rank}.npz"), out=out, rot=rot, lo=lo, hi=hi, keep=keep)
    m1, r1, p1, m2, r2, p2, ang = _grid_case(4)
    ok = fdist.embed_grid_clash_sharded(m1, r1, p1, m2, r2, p2, ang, rank=rank, world=world, allgather_fn=ag,
                                        thresh=1.4, grid_fn=_oracle_grid)
    np.save(os.path.join(out_dir, f"grid_{rank}.npy"), ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scan_and_pose_grid(tmp_path, world):
    _spawn(_worker_parallel_rows, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    base, tors, masks = _chain(16, 3, seed=9)
    angles = o.cartesian_product((0, 120, 240), (0, 180), (0, 120, 240))
    ref_out, ref_rot = o.torsion_scan(base, tors, masks, angles)
    covered = []
    for r in range(world):
        d = np.load(tmp_path / f"scan_{r}.npz")
        lo, hi = int(d["lo"]), int(d["hi"])
        covered.extend(range(lo, hi))
        assert np.array_equal(d["out"], ref_out[lo:hi]) and np.array_equal(d["rot"], ref_rot[lo:hi])
        assert np.array_equal(d["keep"], ref_rot != 0)
    assert covered == list(range(len(angles)))
    m1, r1, p1, m2, r2, p2, ang = _grid_case(4)
    ref_ok, _ = _oracle_grid(m1, r1, p1, m2, r2, p2, ang, thresh=1.4)
    assert 0 < ref_ok.sum() < ref_ok.size
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"grid_{r}.npy"), ref_ok)


def test_shard_bounds_and_mask_gather():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            b = [fdist.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    rng = np.random.default_rng(0)
    full = rng.random((11, 3, 5)) < 0.4
    world = 4
    sent = []
    for r in range(world):
        lo, hi = fdist.shard_bounds(11, r, world)
        fdist.gather_mask_slices(full[lo:hi], 11, r, world, lambda p: sent.append(p) or np.zeros((world, p.shape[0]), np.uint8))
    rows = np.stack(sent)
    for r in range(world):
        lo, hi = fdist.shard_bounds(11, r, world)
        assert np.array_equal(fdist.gather_mask_slices(full[lo:hi], 11, r, world, lambda p: rows), full)
