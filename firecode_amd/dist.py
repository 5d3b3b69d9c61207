"""Conformer-sharded RMSD pruning over the GPUs of one node.

Every rank holds the whole (small) ensemble in its own HBM and owns the row
blocks of the similarity bit matrix dealt in snake order (``owner_of_rows``:
balances the triangular work).  The similarity stage needs no
communication.  The greedy k-ladder needs the *global* survivor mask of the
previous level, so after every level the ranks exchange their rows' new flags
with ONE all-gather of (N,) uint8 (RCCL on GPUs -- ``backend="nccl"`` of
torch.distributed -- gloo in the CPU tests) and combine them: a flag can only
go 1 -> 0 and only its owner changes it, so the element-wise minimum over the
gathered masks is the owner's value.
"""

from __future__ import annotations

import numpy as np

LADDER = (500000, 200000, 100000, 50000, 20000, 10000, 5000, 2000, 1000, 500,
          200, 100, 50, 20, 10, 5, 2, 1)


def owner_of_rows(n, world, row_block):
    """Rank owning each row: row blocks are dealt in snake order (0..W-1, W-1..0,
    ...) because the work of a row block falls linearly with its index -- the
    same map as ``global_block`` in csrc/fc_common.h."""
    block = np.arange(n) // row_block
    cycle, pos = block // world, block % world
    return np.where(cycle % 2 == 0, pos, world - 1 - pos)


def run_ladder(n, level_fn, allgather_fn, min_per_group=20, trace=None):
    """Host control loop shared by the GPU path and the CPU tests.

    level_fn(k, mask_u8) -> this rank's view of the next mask (owned rows
    updated, others copied); allgather_fn(mask_u8) -> (world, n) uint8."""
    mask = np.ones(n, dtype=np.uint8)
    for k in LADDER:
        if k == 1 or min_per_group * k < int(mask.sum()):
            mine = level_fn(k, mask)
            mask = np.ascontiguousarray(allgather_fn(mine).min(axis=0))
            if trace is not None:
                trace.append((k, int(mask.sum())))
    return mask.astype(bool)


def torch_allgather(group=None, device=None):
    """all-gather of a (n,) uint8 mask through torch.distributed (RCCL when the
    process group is nccl and ``device`` is a cuda device, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)

    def fn(mask_u8):
        t = torch.from_numpy(np.ascontiguousarray(mask_u8))
        if device is not None:
            t = t.to(device)
        out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)  # rank-major concatenation
        return out.view(world, t.numel()).cpu().numpy()

    return fn


def prune_by_rmsd_sharded(ens, max_rmsd, max_dev=None, rank=0, world=1, allgather_fn=None,
                          row_block=256, min_per_group=20, trace=None):
    """``ens``: a ``DeviceEnsemble`` holding the whole ensemble on this rank's
    GPU.  Returns (mask (N,) bool, stats of this rank's similarity stage)."""
    if max_dev is None:
        max_dev = 2 * max_rmsd
    if allgather_fn is None:
        if world != 1:
            raise ValueError("allgather_fn is required when world > 1")
        allgather_fn = lambda m: m[None]  # noqa: E731
    stats = ens.prune_begin(max_rmsd, max_dev, rank, world, row_block=row_block)
    mask = run_ladder(ens.N, ens.prune_level, allgather_fn, min_per_group=min_per_group, trace=trace)
    return mask, stats
