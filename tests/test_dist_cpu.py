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

    def similar_pairs(self):
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
    lists = [np.arange(5, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.arange(100, 103, dtype=np.uint64)]
    # emulate 3 ranks in-process: the all-gather stacks what each rank would send
    sent = {}

    def make(rank):
        def fn(buf):
            sent.setdefault(len(buf), {})[rank] = buf.copy()
            return None
        return fn

    # two passes: first record every rank's buffers, then answer from the record
    def run(rank, answers):
        it = iter(answers)
        return fdist.gather_pairs(lists[rank], lambda buf: next(it))

    counts = np.stack([np.array([len(l)], dtype=np.int64).view(np.uint8) for l in lists])
    longest = max(len(l) for l in lists)
    padded = np.stack([np.concatenate([l, np.full(longest - len(l), fdist.PAD, dtype=np.uint64)]).view(np.uint8)
                       for l in lists])
    for rank in range(3):
        out = run(rank, [counts, padded])
        assert np.array_equal(out, np.concatenate(lists))


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
