"""Conformer-sharded RMSD pruning over the GPUs of one node.

Every rank holds the whole (small) ensemble in its own HBM and owns the row
blocks of the similarity bit matrix dealt in snake order (``owner_of_rows``:
balances the triangular work).  The similarity stage needs no
communication.  Then ONE all-gather exchanges the ranks' exactly-similar pair
lists and every rank replays the whole k-ladder (identical mask everywhere);
with dense similarity the ranks exchange the (N,) mask once per ladder level
instead: a flag can only go 1 -> 0 and only its owner changes it, so the
element-wise minimum over the gathered masks is the owner's value.

The exchange is RCCL behind the C ABI (``fc_comm_init``, ``fc_prune_rmsd_sharded``,
``fc_allgather_mask``: csrc/fc_comm.cpp) -- no PyTorch: ``comm_init_from_env``
bootstraps the communicator for one process per GPU started by any launcher that
sets RANK / WORLD_SIZE / LOCAL_RANK (``python -m torch.distributed.run`` does; the
ranks themselves never import torch).  The ``allgather_fn`` / ``gather_fn`` hooks of
the functions below let the CPU tests drive the same host logic over gloo, and
``prune_by_rmsd_sharded_device`` / ``prune_steps_sharded_device`` remain for callers
that already live on torch streams (needs ``FC_HIP_RUNTIME=torch``).
"""

from __future__ import annotations

import os

import numpy as np

LADDER = (500000, 200000, 100000, 50000, 20000, 10000, 5000, 2000, 1000, 500,
          200, 100, 50, 20, 10, 5, 2, 1)


# ---------------------------------------------------------------------------------------
# RCCL without PyTorch: bootstrap of the communicator and the calls over it
# ---------------------------------------------------------------------------------------
def _id_file():
    """Where rank 0 leaves the 128-byte RCCL unique id for the other ranks of this launch:
    FC_COMM_ID_FILE, or a name made of the rendezvous port and the launcher's pid (the ranks of one
    launch share their parent process, so the name is new for every launch)."""
    explicit = os.environ.get("FC_COMM_ID_FILE")
    if explicit:
        return explicit
    if not os.environ.get("MASTER_PORT") and not os.environ.get("TORCHELASTIC_RUN_ID"):
        # srun / mpirun style launches: the ranks do not share a parent, so a name made of the parent's pid
        # would differ from rank to rank and every rank but 0 would wait for the time-out
        raise RuntimeError("several ranks without MASTER_PORT / TORCHELASTIC_RUN_ID: set FC_COMM_ID_FILE (a path every rank "
                           "sees, new for each launch) or FC_COMM_ID (256 hex digits of fc_comm_unique_id)")
    import tempfile

    tag = "_".join([os.environ.get("MASTER_ADDR", "local").replace("/", "_"), os.environ.get("MASTER_PORT", "0"),
                    os.environ.get("TORCHELASTIC_RUN_ID", "none").replace("/", "_"), str(os.getppid())])
    return os.path.join(tempfile.gettempdir(), f"fc_comm_{tag}.id")


class HostRendezvous:
    """Barrier and all-gather of a few bytes between the ranks of ONE NODE through files next to the id file
    (``_id_file()``): what a bench needs to bracket a timed region (barrier, max over ranks) when the measured path
    has no data-path collective -- independent of the RCCL communicator, so that a communicator that cannot be
    created does not take the measurement with it.  Every call is collective and numbered; a rank removes its own
    files two calls later (nobody can still be reading them then)."""

    def __init__(self, rank, world, base=None, timeout_s=None, poll_s=0.0005):
        self.rank, self.world, self.n = int(rank), int(world), 0
        self.base = (base or _id_file()) + ".rv"
        self.timeout_s = float(os.environ.get("FC_COMM_TIMEOUT_S", "180")) if timeout_s is None else float(timeout_s)
        self.poll_s = poll_s
        self._mine = []

    def _path(self, n, rank):
        return f"{self.base}.{n}.{rank}"

    def allgather(self, payload=b""):
        """bytes from every rank, in rank order"""
        import time

        n = self.n
        self.n += 1
        if self.world == 1:
            return [bytes(payload)]
        path = self._path(n, self.rank)
        tmp = f"{path}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(b"\x01" + bytes(payload))  # never empty: a file that exists is complete (rename is atomic)
        os.replace(tmp, path)
        self._mine.append(path)
        out, t0 = [None] * self.world, time.monotonic()
        while any(o is None for o in out):
            for r in range(self.world):
                if out[r] is None:
                    try:
                        with open(self._path(n, r), "rb") as fh:
                            data = fh.read()
                        if data[:1] == b"\x01":
                            out[r] = data[1:]
                    except OSError:
                        pass
            if any(o is None for o in out):
                if time.monotonic() - t0 > self.timeout_s:
                    missing = [r for r in range(self.world) if out[r] is None]
                    raise TimeoutError(f"rank {self.rank}: ranks {missing} did not reach rendezvous {n} within {self.timeout_s:.0f} s")
                time.sleep(self.poll_s)
        while len(self._mine) > 2:  # everybody has passed call n - 1, hence read the files of call n - 2
            try:
                os.remove(self._mine.pop(0))
            except OSError:
                pass
        return out

    def barrier(self):
        self.allgather(b"")

    def max(self, x):
        import struct

        return max(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(x))))

    def close(self):
        """collective, last call: every rank passes one more barrier and says so in a file of its own; rank 0, the last
        reader, waits for those and removes every file of this rendezvous (a rank cannot remove its own last file:
        somebody may still be reading it)"""
        import glob
        import time

        if self.world == 1:
            return
        self.barrier()
        if self.rank != 0:
            path = f"{self.base}.bye.{self.rank}"
            with open(path + ".tmp", "wb") as fh:
                fh.write(b"\x01")
            os.replace(path + ".tmp", path)
            return
        t0 = time.monotonic()
        while not all(os.path.exists(f"{self.base}.bye.{r}") for r in range(1, self.world)):
            if time.monotonic() - t0 > self.timeout_s:
                break  # (a rank that died after the barrier: leave its files, remove the rest)
            time.sleep(self.poll_s)
        for path in glob.glob(glob.escape(self.base) + ".*"):
            try:
                os.remove(path)
            except OSError:
                pass
        self._mine = []


def comm_init_from_env(timeout_s=None):
    """Create the RCCL communicator of this launch from RANK / WORLD_SIZE / LOCAL_RANK and return
    (rank, world, local_rank).  Call it BEFORE any other GPU use in the process; it selects
    device LOCAL_RANK (``fc_init``).  Rank 0 makes the unique id (``fc_comm_unique_id``) and
    publishes it atomically in ``_id_file()``; the others wait for that file.  A single rank needs
    no file.  FC_COMM_ID (256 hex digits) replaces the file for launchers that can pass it."""
    import time

    from firecode_amd import _lib

    if timeout_s is None:
        timeout_s = float(os.environ.get("FC_COMM_TIMEOUT_S", "180"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the host driver supports nothing else
    _lib.init(local_rank)
    hex_id = os.environ.get("FC_COMM_ID")
    if hex_id:
        uid = bytes.fromhex(hex_id)
    elif world == 1:
        uid = _lib.comm_unique_id()
    elif rank == 0:
        uid = _lib.comm_unique_id()
        path = _id_file()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
    else:
        path, t0 = _id_file(), time.monotonic()
        while True:
            try:
                with open(path, "rb") as fh:
                    uid = fh.read()
                if len(uid) == 128:
                    break
            except OSError:
                pass
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no RCCL unique id in {path} after {timeout_s:.0f} s")
            time.sleep(0.02)
        # "rank `rank` has the id": rank 0 removes the id file only when every rank has said so
        with open(f"{path}.ack.{rank}", "wb") as fh:
            fh.write(b"1")
    _lib.comm_init(rank, world, uid)  # collective with RCCL: returns once every rank has joined
    if world > 1 and rank == 0 and not hex_id:
        # ncclCommInitRank is collective, so every rank has read the id by now -- but that is the collective library's
        # property, not this file's: the removal waits for the ranks' own acknowledgements (a stand-in collective whose
        # init returned at once lost the id file under a slow rank, one run in twenty)
        path, t0 = _id_file(), time.monotonic()
        acks = [f"{path}.ack.{r}" for r in range(1, world)]
        while not all(os.path.exists(a) for a in acks):
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank 0: not every rank acknowledged the RCCL unique id in {path} after {timeout_s:.0f} s")
            time.sleep(0.005)
        for f in [path] + acks:
            try:
                os.remove(f)
            except OSError:
                pass
    return rank, world, local_rank


def rccl_allgather():
    """``allgather_fn`` for the host-level functions of this module over the C-ABI communicator:
    (n,) uint8 on every rank -> (world, n) uint8 (``fc_allgather_mask``)."""
    from firecode_amd import _lib

    return _lib.allgather_mask


def prune_by_rmsd_sharded_rccl(ens, max_rmsd, max_dev=None, min_per_group=20, row_block=0):
    """The sharded prune entirely behind the C ABI (``fc_prune_rmsd_sharded``): the ranks of
    ``comm_init_from_env`` call it with the same resident ensemble -> (mask, stats), identical
    mask on every rank.  A group of one without a communicator."""
    if max_dev is None:
        max_dev = 2 * max_rmsd
    return ens.prune_sharded(max_rmsd, max_dev, min_per_group=min_per_group, row_block=row_block)


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


PAD = np.uint64(0xFFFFFFFFFFFFFFFF)  # (i = 2^32-1): ignored by the scatter kernel


NONE = np.uint64(0xFFFFFFFFFFFFFFFE)  # header value of a rank that has no list


def gather_pairs(pairs, allgather_fn, cap=0):
    """Variable-length all-gather of the ranks' uint64 pair lists.

    One collective in the usual case: every rank sends ``cap + 1`` words --
    its count, then its pairs, padded.  Only when some list is longer than
    ``cap`` a second collective (padded to the longest list) follows.
    ``pairs is None`` = "this rank has no list"; then None is returned on every
    rank (the header doubles as the vote for the exchange mode)."""
    n = 0 if pairs is None else int(pairs.shape[0])
    buf = np.full(cap + 1, PAD, dtype=np.uint64)
    buf[0] = NONE if pairs is None else np.uint64(n)
    if n:
        buf[1: 1 + min(n, cap)] = pairs[:cap]
    rows = allgather_fn(buf.view(np.uint8)).view(np.uint64).reshape(-1, cap + 1)
    counts = rows[:, 0]
    if (counts == NONE).any():
        return None
    counts = counts.astype(np.int64)
    if (counts <= cap).all():
        return np.concatenate([rows[r, 1: 1 + counts[r]] for r in range(len(counts))])
    longest = int(counts.max())
    padded = np.full(longest, PAD, dtype=np.uint64)
    padded[:n] = pairs
    full = allgather_fn(padded.view(np.uint8)).view(np.uint64).reshape(len(counts), longest)
    return np.concatenate([full[r, : counts[r]] for r in range(len(counts))])


def exchange_cap(n, world):
    """Pairs a rank can send in the single fixed-size collective (same on every rank)."""
    return 1024 + 4 * n // max(world, 1)


_streams = {}
_buffers = {}
_lane_streams = {}  # device -> [home, second lane, screens]
_lane_buffers = {}  # (device, world, cap) -> [(send, recv)] x 2


def prune_by_rmsd_sharded_device(ens, max_rmsd, max_dev=None, rank=0, world=1, group=None, device=None,
                                 row_block=128, min_per_group=20, gather_fn=None):
    """The sharded prune with the exchange kept in HBM: screen + refine of the own row
    blocks, the rank's similar-pair list written by a kernel into the collective's send
    buffer, ONE ``all_gather_into_tensor`` (RCCL) and the ladder replay are all enqueued
    on one HIP stream -- RCCL orders itself against it -- and the host waits once, for
    the mask.  ``gather_fn(send, recv)`` replaces the collective in single-process tests.

    When some rank's list does not fit the fixed message (or its candidate queue
    overflowed) every rank sees that in the gathered headers and all of them redo the
    exchange through ``prune_by_rmsd_sharded``'s host path together (rare: it repeats
    the similarity stage)."""
    import torch
    import torch.distributed as tdist

    from firecode_amd import _lib

    if max_dev is None:
        max_dev = 2 * max_rmsd
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    stream = _streams.get(device)
    if stream is None:
        stream = _streams[device] = torch.cuda.Stream(device=device)
    cap = exchange_cap(ens.N, world)
    with torch.cuda.stream(stream):
        _lib.stream_set(stream.cuda_stream)
        try:
            key = (device, world, cap)
            if key not in _buffers:  # the collective's buffers are kept: same size every step
                _buffers.clear()
                _buffers[key] = (torch.empty(cap + 1, dtype=torch.int64, device=device),
                                 torch.empty(world * (cap + 1), dtype=torch.int64, device=device))
            send, recv = _buffers[key]
            ens.prune_begin_async(max_rmsd, max_dev, rank, world, row_block=row_block)
            ens.export_pairs_dev(send.data_ptr(), cap)
            if gather_fn is not None:
                gather_fn(send, recv)
            elif world == 1 and group is None and not tdist.is_initialized():
                recv.copy_(send)
            else:
                tdist.all_gather_into_tensor(recv, send, group=group)
            try:
                return ens.prune_from_gathered_dev(recv.data_ptr(), world, cap, min_per_group=min_per_group)
            except _lib.FirecodeHipInputError as e:
                if e.code != _lib.FC_E_LIMIT:  # FC_E_LIMIT comes identically on every rank
                    raise
        finally:
            _lib.stream_set(None)
    allgather = torch_allgather(group=group, device=device) if (world > 1 or tdist.is_initialized()) else None
    return prune_by_rmsd_sharded(ens, max_rmsd, max_dev, rank=rank, world=world, allgather_fn=allgather,
                                 row_block=row_block, min_per_group=min_per_group)


def prune_steps_sharded_device(ens, steps, max_rmsd, max_dev=None, rank=0, world=1, group=None, device=None,
                               row_block=128, min_per_group=20, gather_fn=None, overlap=None):
    """``steps`` sharded prunes of the same resident ensemble, STREAM-ORDERED: screen + refine,
    export, the RCCL all-gather and the ladder of every prune are enqueued without a host wait in
    between and the host waits once at the end (``fc_prune_collect``).

    ``overlap`` (default: on for more than one step, ``FC_SHARD_LANES=1`` turns it off): all
    screens go, in order, to one stream; counters reset, refine, export, all-gather and ladder of
    prune k go to stream k&1 of two others and work on workspace k&1 (the ensemble and its twin,
    each with its own send / receive buffers), so they run beside the screen of prune k + 1.
    Prune k + 2 is ordered behind prune k on the same stream.  torch issues every collective on
    the process group's own stream in call order, so the all-gathers of the two lanes stay
    serialised and in the same order on every rank.  Without overlap everything is enqueued on
    one stream.  Returns a list of (mask, stats); a prune whose device ladder declined is redone
    through the host exchange on every rank alike."""
    import torch
    import torch.distributed as tdist

    from firecode_amd import _lib

    if max_dev is None:
        max_dev = 2 * max_rmsd
    if overlap is None:
        overlap = os.environ.get("FC_SHARD_LANES") != "1"
    overlap = bool(overlap) and steps > 1
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    pool = _lane_streams.get(device)
    if pool is None:
        # the screen fills every workgroup slot of the chip: the lanes' small kernels (and RCCL behind them)
        # need the dispatcher's preference to run beside it, so the lanes get the highest stream priority
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        pool = _lane_streams[device] = [torch.cuda.Stream(device=device, priority=hi),
                                        torch.cuda.Stream(device=device, priority=hi),
                                        torch.cuda.Stream(device=device, priority=lo)]
    home, lane_b, screen = pool
    lanes = [home, lane_b] if overlap else [home, home]
    work = [ens, ens.twin()] if overlap else [ens, ens]
    cap = exchange_cap(ens.N, world)
    key = (device, world, cap)
    if key not in _lane_buffers:
        _lane_buffers.clear()
        _lane_buffers[key] = [(torch.empty(cap + 1, dtype=torch.int64, device=device),
                          torch.empty(world * (cap + 1), dtype=torch.int64, device=device)) for _ in range(2)]
    results = []

    def gather(send, recv):
        if gather_fn is not None:
            gather_fn(send, recv)
        elif world == 1 and group is None and not tdist.is_initialized():
            recv.copy_(send)
        else:
            tdist.all_gather_into_tensor(recv, send, group=group)

    caller = torch.cuda.current_stream(device)
    home.wait_stream(caller)
    _lib.stream_set(home.cuda_stream)
    try:
        if overlap:  # both other streams start behind what the home stream holds
            lane_b.wait_stream(home)
            screen.wait_stream(home)
        for k in range(steps):
            e, lane = work[k & 1], lanes[k & 1]
            send, recv = _lane_buffers[key][k & 1]
            with torch.cuda.stream(lane):
                _lib.stream_use(lane.cuda_stream)
                if overlap:
                    e.prune_begin_split_async(max_rmsd, max_dev, rank, world, screen.cuda_stream, row_block=row_block,
                                              timed=(k % 8 == 0 or k == steps - 1))  # stats[4]: the last timed screen
                else:
                    e.prune_begin_async(max_rmsd, max_dev, rank, world, row_block=row_block)
                e.export_pairs_dev(send.data_ptr(), cap)
                gather(send, recv)
                e.prune_from_gathered_enqueue(recv.data_ptr(), world, cap, k, steps, min_per_group=min_per_group)
        _lib.stream_use(home.cuda_stream)
        if overlap:
            home.wait_stream(lane_b)
            home.wait_stream(screen)
        for k in range(steps):
            try:
                results.append(work[k & 1].prune_collect(k, steps))  # the first one waits for the home stream
            except _lib.FirecodeHipInputError as e:
                if e.code != _lib.FC_E_LIMIT:
                    raise
                results.append(None)
    finally:
        _lib.stream_use(home.cuda_stream)
        _lib.stream_set(None)
    if any(r is None for r in results):
        allgather = torch_allgather(group=group, device=device) if (world > 1 or tdist.is_initialized()) else None
        results = [r if r is not None else
                   prune_by_rmsd_sharded(ens, max_rmsd, max_dev, rank=rank, world=world, allgather_fn=allgather,
                                         row_block=row_block, min_per_group=min_per_group) for r in results]
    return results


def prune_by_rmsd_sharded(ens, max_rmsd, max_dev=None, rank=0, world=1, allgather_fn=None,
                          row_block=128, min_per_group=20, trace=None, mode="auto"):
    """``ens``: a ``DeviceEnsemble`` holding the whole ensemble on this rank's
    GPU.  Each rank computes the similarity of its own row blocks, then

    * ``pairs`` (default when available): ONE variable-length all-gather of the
      ranks' exactly-similar pair lists (a few hundred kB), after which every
      rank replays the whole k-ladder locally -- no further communication;
    * ``levels``: one (N,) uint8 mask all-gather per ladder level (used when a
      rank's candidate queue overflowed, i.e. similarity is dense).

    Returns (mask (N,) bool, stats of this rank's similarity stage)."""
    if max_dev is None:
        max_dev = 2 * max_rmsd
    if allgather_fn is None:
        if world != 1:
            raise ValueError("allgather_fn is required when world > 1")
        allgather_fn = lambda m: m[None]  # noqa: E731
    stats = ens.prune_begin(max_rmsd, max_dev, rank, world, row_block=row_block)
    pairs = None
    if mode in ("auto", "pairs") and hasattr(ens, "similar_pairs"):
        try:
            pairs = ens.similar_pairs(int(stats[2]))
        except ValueError:  # FirecodeHipInputError(FC_E_LIMIT): queue overflow on this rank
            if mode == "pairs":
                raise
    # every rank takes the same branch: the count header doubles as the vote; the
    # capacity (same on every rank) makes the exchange ONE collective unless similarity is
    # much denser than a handful of duplicates per conformer
    all_pairs = gather_pairs(pairs, allgather_fn, cap=exchange_cap(ens.N, world))
    if all_pairs is not None:
        mask = ens.prune_from_pairs(all_pairs, min_per_group=min_per_group)
        return mask, stats
    mask = run_ladder(ens.N, ens.prune_level, allgather_fn, min_per_group=min_per_group, trace=trace)
    return mask, stats


# ---------------------------------------------------------------------------------------
# The embarrassingly parallel rows: clash / embed / torsion scan.
# A contiguous slice of the outermost index per rank, the compute call unchanged, then ONE
# all-gather of the bit-packed pass mask -- no other collective.
# ---------------------------------------------------------------------------------------
def shard_bounds(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of range(n) for ``rank`` (the first n % world
    ranks get one more)."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_mask_slices(local_mask, n_total, rank, world, allgather_fn):
    """All-gather of per-rank boolean slices (leading axis sharded by ``shard_bounds``):
    bits are packed 8 per byte, padded to the longest slice, exchanged with ONE collective
    and reassembled in rank order.  Returns the full (n_total, ...) boolean array."""
    local_mask = np.ascontiguousarray(local_mask, dtype=bool)
    inner = local_mask.shape[1:]
    per_row = int(np.prod(inner)) if inner else 1
    longest = shard_bounds(n_total, 0, world)[1] * per_row  # rank 0 holds the longest slice
    packed = np.zeros((longest + 7) // 8, dtype=np.uint8)
    mine = np.packbits(local_mask.reshape(-1))
    packed[: mine.shape[0]] = mine
    rows = np.asarray(allgather_fn(packed)).reshape(world, -1)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        nbits = (hi - lo) * per_row
        parts.append(np.unpackbits(rows[r])[:nbits].astype(bool).reshape((hi - lo,) + inner))
    return np.concatenate(parts, axis=0)


def torsion_scan_sharded(base, torsions, masks, angles, rank=0, world=1, allgather_fn=None, thresh=1.5,
                         scan_fn=None):
    """Angle-sets sharded over the ranks (clustered_csearch's inner loops,
    torsion_module.py:812-856).  Every rank scans its slice on its GPU and keeps those
    conformers; the (S,) "at least one bond rotated" flags are all-gathered.
    Returns (local_coords (s_local, A, 3), local_rotated, (lo, hi), keep_all (S,) bool)."""
    if scan_fn is None:
        from firecode_amd.torsion_module import torsion_scan as scan_fn
    if allgather_fn is None:
        allgather_fn = lambda m: m[None]  # noqa: E731
    angles = np.asarray(angles)
    lo, hi = shard_bounds(len(angles), rank, world)
    out, rot = scan_fn(base, torsions, masks, angles[lo:hi], thresh=thresh)
    keep_all = gather_mask_slices(np.asarray(rot) != 0, len(angles), rank, world, allgather_fn)
    return out, rot, (lo, hi), keep_all


def embed_grid_clash_sharded(m1, reactive1, pivots1, m2, reactive2, pivots2, angles1, angles2=None, rank=0,
                             world=1, allgather_fn=None, thresh=1.5, max_clashes=0, grid_fn=None):
    """The bimolecular pose grid (embeds.py:597-722) sharded over the conformers of the SECOND
    molecule -- the slowest index of the reference's loop, so every rank owns a contiguous
    block of the flattened pose order.  One all-gather of the packed pass mask.
    Returns pass (n2, n1, 2, na2, na1) bool, identical on every rank."""
    if grid_fn is None:
        from firecode_amd.embeds import embed_grid_clash as grid_fn
    if allgather_fn is None:
        allgather_fn = lambda m: m[None]  # noqa: E731
    m2 = np.asarray(m2)
    lo, hi = shard_bounds(len(m2), rank, world)
    if hi > lo:
        local = grid_fn(m1, reactive1, pivots1, m2[lo:hi], reactive2, np.asarray(pivots2)[lo:hi], angles1, angles2,
                        thresh=thresh, max_clashes=max_clashes)[0]
    else:
        a1 = np.asarray(angles1).reshape(-1)
        a2 = a1 if angles2 is None else np.asarray(angles2).reshape(-1)
        local = np.zeros((0, len(np.asarray(m1)), 2, len(a2), len(a1)), dtype=bool)
    return gather_mask_slices(local, len(m2), rank, world, allgather_fn)
