"""World > 1 through the C path on ONE device.  RCCL refuses a communicator whose ranks share a GPU, and the pool gives a
test one GPU, so the collectives of fc_comm.cpp had only ever run in a 1-rank communicator.  Here two and three real rank
PROCESSES (started fresh, before any GPU call) load tests/stubs/rccl_stub.cpp through FC_RCCL_LIB -- a stand-in that
accepts the duplicate device and moves the all-gather's bytes through shared memory -- and run fc_comm_init ->
fc_prune_rmsd_sharded / fc_bench_prune_rmsd_sharded (both lanes, the s_comm event ordering, staging buffers) and the
per-level mask all-gather of the dense-similarity fallback end to end.  Every rank's mask must equal the single-GPU mask.

This is a correctness rehearsal, NOT a scaling measurement: no scaling curve has been measured until an 8-GPU node runs
`bench.py --gpus 8` (README / DESIGN section 7)."""

import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib, dist as fdist, synthetic as syn
rank, world, _ = fdist.comm_init_from_env()
assert _lib.comm_info() == (rank, world)
X, atoms, asg = syn.synthetic_ensemble({n}, 30, seed=21)
with fc.DeviceEnsemble(X, center=True) as ens:
    ref, st0 = ens.prune(0.5, 1.0)                      # this rank alone, every row: the single-GPU answer
    mask, st = ens.prune_sharded(0.5, 1.0)              # rows dealt to the ranks, one all-gather, ladder replayed
    _, _, mask_lanes, st_l = ens.bench_prune_sharded(0.5, 1.0, reps=5, overlap=True)   # two workspaces / lanes
    _, _, mask_serial, _ = ens.bench_prune_sharded(0.5, 1.0, reps=3, overlap=False)
gathered = _lib.allgather_mask(np.full(7, rank + 1, dtype=np.uint8))
assert gathered.shape == (world, 7) and all((gathered[r] == r + 1).all() for r in range(world))
np.savez({out!r} + str(rank) + ".npz", ref=ref, mask=mask, mask_lanes=mask_lanes, mask_serial=mask_serial,
         owned=int(st[0]), similar_local=int(st[2]), similar_all=int(st0[2]), clusters=len(np.unique(asg)))
_lib.comm_barrier()
_lib.comm_destroy()
print("rank", rank, "ok")
"""


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is needed to build the RCCL stand-in")
    out = str(tmp_path_factory.mktemp("stub") / "librccl_stub.so")
    src = os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", src, "-o", out,
                        "-lrt", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.mark.parametrize("world,n,dense", [(2, 2400, False), (3, 1500, False), (2, 900, True)])
def test_sharded_prune_across_rank_processes_on_one_device(stub_lib, tmp_path, world, n, dense):
    idfile = str(tmp_path / "comm.id")
    out = str(tmp_path / "rank")
    code = CHILD.format(root=ROOT, n=n, out=out)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", FC_COMM_ID_FILE=idfile,
                   FC_RCCL_LIB=stub_lib, FC_COMM_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if dense:
            env["FC_PAIRQ_CAP"] = "4"  # the pair queue overflows: similarity bits + one mask all-gather per ladder level
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise
    bad = [f"rank {rank} (exit {p.returncode}): {se[-1500:]}" for rank, (p, (so, se)) in enumerate(zip(procs, outs))
           if p.returncode != 0 or f"rank {rank} ok" not in so]
    assert not bad, "\n".join(bad)  # (every rank's end: the first one to fail is usually not the one that caused it)
    res = [np.load(out + f"{r}.npz") for r in range(world)]
    ref = res[0]["ref"]
    assert 0 < ref.sum() == int(res[0]["clusters"]) < n
    for r in range(world):
        for key in ("ref", "mask", "mask_lanes", "mask_serial"):
            assert np.array_equal(res[r][key], ref), (r, key)
    pairs_total = n * (n - 1) // 2
    assert sum(int(r["owned"]) for r in res) == pairs_total and all(int(r["owned"]) > 0 for r in res)
    if not dense:  # every similar pair is found by exactly one rank
        assert sum(int(r["similar_local"]) for r in res) == int(res[0]["similar_all"])


CHILD_VERDICT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import firecode_amd as fc
from firecode_amd import _lib, dist as fdist
rank, world, _ = fdist.comm_init_from_env()
rng = np.random.default_rng(77)
bases = rng.normal(scale=2.5, size=(128, 1, 20, 3))                     # 128 shapes x 8 noisy copies: similar pairs only inside a shape
X = (bases + rng.normal(scale=0.05, size=(128, 8, 20, 3))).reshape(1024, 20, 3)
X[128:384] = bases[16] + rng.normal(scale=0.4, size=(256, 20, 3))      # row blocks 1 and 2 (rank 1's): a family ~1 A apart, none similar
X = X + np.array([40.0, -25.0, 10.0])                                   # uncentred, far from the origin: a wide undecidable band
expected = np.zeros(1024, dtype=bool)
expected[7::8] = True                                                   # the last copy of every shape survives ...
expected[128:384] = True                                                # ... and every member of the family
with fc.DeviceEnsemble(X, center=False) as ens:
    ref, _ = ens.prune(0.5, 1.0)                                         # this rank alone (FC_SCREEN_F32=3: verdict, fp64 redo in place)
    assert np.array_equal(ref, expected)
    mask, st = ens.prune_sharded(0.5, 1.0)
    print("rank", rank, "pair path" if st[4] else "per-level path", flush=True)
    _, _, mask_lanes, _ = ens.bench_prune_sharded(0.5, 1.0, reps=3, overlap=True)
np.savez({out!r} + str(rank) + ".npz", ref=ref, mask=mask, mask_lanes=mask_lanes)
_lib.comm_barrier()
_lib.comm_destroy()
print("rank", rank, "ok")
"""


def test_a_verdict_on_one_rank_only(stub_lib, tmp_path):
    """The speculative single-precision screen is voted down by a kernel that samples the RANK'S OWN candidate queue.  Here
    only rank 1's rows hold pairs inside the undecidable band (a family of 256 conformers ~1 A apart in row blocks 1 and 2,
    everything else well-separated clusters; uncentred coordinates 48 A from the origin make the band ~2 A^2 wide;
    FC_SCREEN_F32=3 takes the speculative screen whatever the band): rank 1 is voted down and redoes its rows with the
    fp64 screen IN PLACE (the sharded pipeline does not use the single-GPU pipeline's "decline and redo later" mode),
    rank 0 is not -- both stay on the pair path, one all-gather each, and every rank's mask is the known answer.  (Should
    a declined screen ever reach the exchange, its message says "none" and every rank declines together: k_export_pairs.)"""
    idfile = str(tmp_path / "comm.id")
    out = str(tmp_path / "rank")
    code = CHILD_VERDICT.format(root=ROOT, out=out)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", FC_COMM_ID_FILE=idfile, FC_RCCL_LIB=stub_lib,
                   FC_COMM_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env["FC_SCREEN_F32"] = "3"  # the speculative screen with the verdict behind it, whatever the band (read once per process)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise
    bad = [f"rank {rank} (exit {p.returncode}): {se[-1500:]}" for rank, (p, (so, se)) in enumerate(zip(procs, outs))
           if p.returncode != 0 or f"rank {rank} ok" not in so]
    assert not bad, "\n".join(bad)
    assert all("pair path" in so for so, _ in outs), [so for so, _ in outs]  # nobody fell back to the per-level exchange
    res = [np.load(out + f"{r}.npz") for r in range(2)]
    ref = res[0]["ref"]
    assert ref.sum() == 96 + 256  # one survivor per remaining shape (the family replaced the copies of 32 shapes), every member of the family
    for r in range(2):
        for key in ("ref", "mask", "mask_lanes"):
            assert np.array_equal(res[r][key], ref), (r, key)


def test_bench_two_ranks_with_a_working_communicator(stub_lib):
    """`python bench.py --gpus 2` as typed, both ranks on device 0 (FC_BENCH_SAME_DEVICE=1) with the stand-in collective:
    the blocks that exchange data (prune_path, cfg4_family) run for real -- masks checked against the synthetic
    ensembles' known answers -- and the line says two ranks were seen.  The timings mean nothing (two ranks share a GPU,
    the stand-in blocks the host); only the code path is under test."""
    import json

    env = dict(os.environ, FC_BENCH_SAME_DEVICE="1", FC_RCCL_LIB=stub_lib, FC_BENCH_EXTRAS_TIMEOUT_S="600",
               FC_BENCH_SETTLE_S="0.01")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value_check"]["ok"] and line["value_check"]["all_ranks_ok"]
    assert "RCCL communicator up" in line["rank_coordination"] and "extras_error" not in line
    for blk in (line["prune_path"], line["cfg4_family"]):
        assert blk["survivor_count_ok"] and blk["survivors_are_last_cluster_members"], blk
        assert "all-gather" in blk["sharding"]


def test_bench_four_ranks_rehearsal_and_failing_ranks(stub_lib):
    """The launch the driver makes on an 8-GPU node, rehearsed as far as a one-GPU box allows (at most six processes may
    use its card): `python bench.py --gpus 4` as typed, every rank on device 0, stand-in collective, conformer counts of
    every family scaled by 0.3 (FC_BENCH_REHEARSAL_SCALE: stated in the line).  Four ranks seen, efficiency present, the
    exchanging blocks checked against their known answers.  Then a rank that raises BEFORE and one that raises AFTER the
    communicator is up: the launcher exits non-zero either way and ends the other ranks."""
    import json

    env = dict(os.environ, FC_BENCH_SAME_DEVICE="1", FC_RCCL_LIB=stub_lib, FC_BENCH_EXTRAS_TIMEOUT_S="600",
               FC_BENCH_SETTLE_S="0.01", FC_BENCH_REHEARSAL_SCALE="0.3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["ranks_seen"] == 4 and "efficiency" in line
    assert line["config"]["rehearsal_scale"] == 0.3
    assert line["value_check"]["ok"] and line["value_check"]["all_ranks_ok"]
    assert "RCCL communicator up" in line["rank_coordination"] and "extras_error" not in line
    for blk in (line["prune_path"], line["cfg4_family"]):
        assert blk["survivor_count_ok"], blk
    for spec in ("2:before", "1:after"):
        r = subprocess.run(cmd, env=dict(env, FC_BENCH_FAIL_RANK=spec, FC_COMM_TIMEOUT_S="20"), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0, spec
        assert "FC_BENCH_FAIL_RANK" in r.stderr, r.stderr[-2000:]
