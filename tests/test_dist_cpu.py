"""world_size-2 (and 3) gloo tests of the sharded ladder: the host control
loop, row ownership and the mask exchange of firecode_amd.dist, with the
per-level GPU kernels replaced by an oracle-backed stand-in (CPU only)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, a, seed, row_block, out_dir, pairs_mode=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
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
    mp.spawn(_worker, args=(world, _free_port(), n, a, seed, row_block, str(tmp_path)), nprocs=world, join=True)
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
    mp.spawn(_worker, args=(world, _free_port(), n, a, seed, row_block, str(tmp_path), True), nprocs=world, join=True)
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
