"""The library next to PyTorch in one process (FC_HIP_RUNTIME=torch): callers that already live on
torch streams hand the library their stream and torch tensors as the collective's buffers
(firecode_amd.dist.prune_by_rmsd_sharded_device / prune_steps_sharded_device, RCCL through
torch.distributed).  The product's default is the SYSTEM HIP runtime and no torch, so these tests
run in a child interpreter with the opt-in set before the library is loaded."""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from firecode_amd import synthetic as syn  # noqa: E402
from oracle import cpu_ref as o  # noqa: E402

pytestmark = pytest.mark.gpu
CHILD = os.environ.get("FC_INTEROP_CHILD") == "1"


@pytest.mark.skipif(CHILD, reason="this is the child")
def test_torch_interop_suite_in_a_child_interpreter():
    env = dict(os.environ, FC_HIP_RUNTIME="torch", FC_INTEROP_CHILD="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
                          "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "skipped" in out.stdout  # the children ran, only this launcher skipped


needs_child = pytest.mark.skipif(not CHILD, reason="runs in the child interpreter started above")


@needs_child
def test_runtime_is_shared_with_torch(fc):
    from firecode_amd import _lib

    assert _lib.HIP_RUNTIME == "torch"
    import torch

    assert torch.cuda.is_available()


@needs_child
@pytest.mark.parametrize("world", [1, 3])
def test_device_resident_exchange_logical_ranks(fc, world):
    """fc_prune_export_pairs_dev / fc_prune_from_gathered_dev: the messages of `world`
    logical ranks are written by kernels into one torch buffer (standing in for the
    all-gather's output), the ladder is replayed from it -- all on a torch stream"""
    import torch

    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1100, 24, seed=180 + world)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    cap = fdist.exchange_cap(len(X), world)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with fc.DeviceEnsemble(X, center=True) as ens, torch.cuda.stream(stream):
        _lib.stream_set(stream.cuda_stream)
        try:
            recv = torch.zeros(world * (cap + 1), dtype=torch.int64, device=dev)
            for r in range(world):
                ens.prune_begin_async(0.5, 1.0, r, world, row_block=128)
                ens.export_pairs_dev(recv.data_ptr() + 8 * r * (cap + 1), cap)
            mask, stats = ens.prune_from_gathered_dev(recv.data_ptr(), world, cap)
            host = recv.cpu().numpy().view(np.uint64).reshape(world, cap + 1)
        finally:
            _lib.stream_set(None)
    assert np.array_equal(mask, ref)
    assert int(host[:, 0].sum()) == int(np.triu(S0, 1).sum())
    for r in range(world):  # message layout: count, pairs, padding
        c = int(host[r, 0])
        assert (host[r, 1 + c:] == fdist.PAD).all() and (host[r, 1: 1 + c] != fdist.PAD).all()
    assert stats[5] == ref.sum() and stats[4] > 0


@needs_child
def test_device_resident_exchange_driver_and_fallback(fc, monkeypatch):
    """prune_by_rmsd_sharded_device on one rank (copy instead of the collective); then with a
    4-entry candidate queue: the message header says "no list", the device ladder declines
    (FC_E_LIMIT) and the driver repeats the exchange on the host path -- same mask"""
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(700, 20, seed=191)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    with fc.DeviceEnsemble(X, center=True) as ens:
        mask, stats = fdist.prune_by_rmsd_sharded_device(ens, 0.5)
        assert np.array_equal(mask, ref) and stats[2] == np.triu(S0, 1).sum()
    monkeypatch.setenv("FC_PAIRQ_CAP", "4")
    with fc.DeviceEnsemble(X, center=True) as ens:
        mask, stats = fdist.prune_by_rmsd_sharded_device(ens, 0.5)
        assert np.array_equal(mask, ref)
    # a message longer than the fixed capacity takes the same way out
    monkeypatch.delenv("FC_PAIRQ_CAP")
    monkeypatch.setattr(fdist, "exchange_cap", lambda n, world: 16)
    with fc.DeviceEnsemble(X, center=True) as ens:
        mask, stats = fdist.prune_by_rmsd_sharded_device(ens, 0.5)
        assert np.array_equal(mask, ref)


@needs_child
def test_stream_ordered_sharded_steps(fc, monkeypatch):
    """prune_steps_sharded_device: several prunes enqueued back to back, one host wait; every
    prune delivers the right mask; a declined device ladder is redone through the host path"""
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(900, 22, seed=197)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    with fc.DeviceEnsemble(X, center=True) as ens:
        res = fdist.prune_steps_sharded_device(ens, 5, 0.5)
        assert len(res) == 5
        for mask, stats in res:
            assert np.array_equal(mask, ref) and stats[2] == np.triu(S0, 1).sum() and stats[5] == ref.sum()
    monkeypatch.setattr(fdist, "exchange_cap", lambda n, world: 16)  # messages longer than the capacity
    with fc.DeviceEnsemble(X, center=True) as ens:
        for mask, stats in fdist.prune_steps_sharded_device(ens, 3, 0.5):
            assert np.array_equal(mask, ref)


@needs_child
@pytest.mark.parametrize("overlap", [False, True])
def test_sharded_steps_two_logical_ranks_with_and_without_overlap(fc, overlap):
    """rank 0 of a 2-rank prune on one GPU: rank 1's message is computed beforehand and the
    collective is replaced by a copy of both messages; consecutive steps on one stream, or
    overlapped (ensemble + twin workspace, screens on a stream of their own) -- same masks"""
    import torch

    from firecode_amd import _lib
    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1100, 18, seed=199)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    world, cap = 2, fdist.exchange_cap(len(X), 2)
    with fc.DeviceEnsemble(X, center=True) as ens:
        other = torch.empty(cap + 1, dtype=torch.int64, device="cuda:0")
        ens.prune_begin_async(0.5, 1.0, 1, world)
        ens.export_pairs_dev(other.data_ptr(), cap)
        _lib.call("fc_memory_trim")  # also a host wait for the library's stream
        calls = []

        def gather(send, recv):
            calls.append(send.data_ptr())
            recv.view(world, cap + 1)[0].copy_(send)
            recv.view(world, cap + 1)[1].copy_(other)

        res = fdist.prune_steps_sharded_device(ens, 6, 0.5, rank=0, world=world, gather_fn=gather, overlap=overlap)
        assert len(res) == 6 and len(calls) == 6
        own = int(np.triu(S0, 1)[fdist.owner_of_rows(len(X), world, 128) == 0].sum())
        for mask, stats in res:
            assert np.array_equal(mask, ref) and stats[2] == own and stats[5] == ref.sum()
        # the plain one-shot call still works on the same ensemble afterwards
        mask, _ = ens.prune(0.5, 1.0)
        assert np.array_equal(mask, ref)


@needs_child
def test_device_resident_exchange_over_rccl_single_rank(fc):
    """the real collective (torch.distributed nccl = RCCL) on a 1-rank group: stream ordering
    between the library's kernels and RCCL's stream, no host synchronisation in between"""
    import torch
    import torch.distributed as tdist

    from firecode_amd import dist as fdist

    X, atoms, asg = syn.synthetic_ensemble(1500, 30, seed=195)
    S0, _, _ = o.rmsd_similarity_matrix(X, atoms, 0.5)
    ref = o.greedy_prune_from_matrix(S0)
    torch.cuda.set_device(0)
    tdist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1,
                             device_id=torch.device("cuda", 0))
    try:
        with fc.DeviceEnsemble(X, center=True) as ens:
            for _ in range(3):
                mask, stats = fdist.prune_by_rmsd_sharded_device(ens, 0.5, rank=0, world=1)
                assert np.array_equal(mask, ref)
            for mask, stats in fdist.prune_steps_sharded_device(ens, 4, 0.5, rank=0, world=1):
                assert np.array_equal(mask, ref)
    finally:
        tdist.destroy_process_group()


