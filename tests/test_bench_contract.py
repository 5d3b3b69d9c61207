"""bench.py prints ONE JSON line with the contract's fields: the top-level fields are the fp64
complete-alignment measurement (SURVEY 8d's unit), the pruning step rides as `prune_path`; the same on
the multi-GPU code path (RCCL communicator of one rank through the C ABI, whose stdout RCCL would otherwise
share) and through the self-spawn path `python bench.py --gpus N` takes; the ranks never import torch."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(extra_env, args, expect_ok=True):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    if not expect_ok:
        return out
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_headline(d, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == "conformer-pair RMSD alignments/s" and d["unit"] == "alignments/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    # the headline is the reference-precision measurement: complete fp64 alignments, fp64 peak
    assert d["dtype"] == "f64"
    r = d["roofline"]
    assert r["dtype"] == "f64" and r["peak"] == 78.6 and r["kernel"] == "k_simbits_screen_mfma<4, 2, 64, true>"
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1.5 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["flops_per_alignment"] == 53 * 50 + 600
    assert r["traffic"] is None or r["traffic_source"].startswith("from_file")
    c = d["config"]
    assert c["baseline_config"] == "configs[1]" and "complete alignments" in c["value_counts"] and "workload" in c
    assert c["pairs_per_step"] == 10000 * 9999 // 2 == c["pairs_per_step_rank0"]
    assert abs(d["value"] - c["pairs_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the kernel's event time cannot exceed the step it is part of
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02
    assert d["roofline_hbm"]["bound"] == "hbm" and d["roofline_hbm"]["peak"] == 8000.0


def _check_prune_block(p):
    assert (p["dtype"], p["roofline"]["dtype"], p["roofline"]["peak"]) in (
        ("f16x2 screen (split-half, fp32-accurate) + f64 exact refine", "f16x2", 2500.0),
        ("f32 screen + f64 exact refine", "f32", 157.3), ("f64", "f64", 78.6))
    assert p["survivor_count_ok"] is True and p["survivors_are_last_cluster_members"] is True
    assert p["pair_decisions_per_s"] > 0 and p["pruned_ensembles_per_s"] > 0 and 0 < p["roofline"]["frac"] < 1
    assert abs(p["pair_decisions_per_s"] - p["pairs_per_step"] / (p["ms_per_step"] * 1e-3)) / p["pair_decisions_per_s"] < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True])
def test_bench_prints_one_contract_line(sharded):
    d = _run({"FC_BENCH_FORCE_SHARDED": "1"} if sharded else {}, ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    _check_headline(d, 2, 1)
    _check_prune_block(d["prune_path"])
    assert d["pruned_ensembles_per_s"] == d["prune_path"]["pruned_ensembles_per_s"]
    if sharded:
        assert d["prune_path"]["comm"].startswith("RCCL through libfc_hip.so") and d["ranks_seen"] == 1
    # the other readings ride in the same line, each with its own device timing and roofline
    f = d["fp64_path"]
    assert f["dtype"] == "f64" and f["roofline"]["peak"] == 78.6 and 0 < f["roofline"]["frac"] < 1
    assert f["mask_equals_default_path"] is True and f["pair_decisions_per_s"] > 0
    assert d["rmsd_values_per_s"] > 0 and d["config"]["secondary"]["ms_per_step"] > 0
    assert d["pruned_ensembles_per_s_host_in_mask_out"] > 0


@pytest.mark.gpu
def test_bench_cpu_baseline_block():
    d = _run({"FC_BENCH_CPU_SECONDS": "3"}, ["--steps", "1", "--warmup", "0", "--no-extras"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "alignments/s" and c["value"] > 0 and c["sample"]
    assert abs(d["vs_cpu_baseline"] - d["value"] / c["value"]) < 1e-6 * d["vs_cpu_baseline"] and "prune_path" not in d


@pytest.mark.gpu
def test_bench_cfg4_and_cfg5_workloads_run_on_one_rank():
    d = _run({}, ["--workload", "cfg4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["config"]["baseline_config"] == "configs[3]" and d["config"]["n_atoms"] == 80 and d["unit"] == "pair decisions/s"
    _check_prune_block(dict(d["prune_path"], roofline=d["roofline"]))
    d = _run({}, ["--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert d["config"]["baseline_config"] == "configs[4]" and d["unit"] == "poses/s" and d["value"] > 0


@pytest.mark.gpu
def test_bench_self_spawn_world_one_and_missing_device():
    """`python bench.py --gpus N` as typed: the parent starts the ranks itself.  World 1 through that path gives the
    same line; --gpus 2 on a one-GPU box fails IN THE CHILD with a device message, not with a usage string."""
    d = _run({"FC_BENCH_FORCE_SPAWN": "1"}, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    _check_headline(d, 2, 1)
    assert d["ranks_seen"] == 1
    from firecode_amd import _lib

    if _lib.device_count() == 1:
        out = _run({"FC_COMM_TIMEOUT_S": "20"}, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"],
                   expect_ok=False)
        assert out.returncode != 0 and out.stdout.strip() == ""
        assert "rank 1 exited" in out.stderr and "device" in out.stderr.lower() and "launch with" not in out.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_agree_through_files(tmp_path):
    """Two bench ranks started the way a launcher starts them (RANK / WORLD_SIZE / id file), both on device 0: the id file
    rendezvous and ncclCommInitRank run for real; RCCL refuses two ranks on one device, so the ranks fall back to the file
    rendezvous for barrier and max -- the headline path has no exchange -- shard the pair matrix by logical rank and rank 0
    prints the one line (or RCCL accepts and the line says so)."""
    idfile = str(tmp_path / "two.id")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                   FC_COMM_ID_FILE=idfile, FC_COMM_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                                       "--no-cpu-baseline", "--no-extras"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs[0][1][-1500:] + outs[1][1][-1500:]
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and outs[1][0].strip() == ""
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["n_conformers"] == 14142 and 0 < d["config"]["pairs_per_step_rank0"] < d["config"]["pairs_per_step"]
    assert d["rank_coordination"].startswith("barrier / max over ranks through files") and 0 < d["efficiency"] < 1.2
    assert not list(tmp_path.glob("two.id.rv.*"))  # the rendezvous cleaned up after itself


@pytest.mark.gpu
def test_bench_gpus_3_as_typed_on_one_device():
    """`python bench.py --gpus 3` as the driver types it, rehearsed on one device (FC_BENCH_SAME_DEVICE=1): the parent starts
    three rank processes, relays rank 0's single line, exits 0; the ranks agree through files (RCCL refuses a duplicate device)"""
    d = _run({"FC_BENCH_SAME_DEVICE": "1", "FC_COMM_TIMEOUT_S": "120"},
             ["--gpus", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["n_conformers"] == 17321 and 0 < d["efficiency"] < 1.2
    assert "scaling_family_n1" in d and d["rank_coordination"].startswith("barrier / max over ranks through files")


def test_host_rendezvous_three_processes(tmp_path):
    """firecode_amd.dist.HostRendezvous: all-gather / max / barrier between three processes, files removed at the end"""
    code = (
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from firecode_amd.dist import HostRendezvous\n"
        f"r = int(sys.argv[1]); rv = HostRendezvous(r, 3, base={str(tmp_path / 'rv')!r}, timeout_s=60)\n"
        "for i in range(40):\n"
        "    g = rv.allgather(bytes([r, i]))\n"
        "    assert [x[0] for x in g] == [0, 1, 2] and all(x[1] == i for x in g)\n"
        "assert rv.max(1.5 * r) == 3.0\n"
        "rv.barrier(); rv.close(); print('ok', r)\n")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, text=True) for r in range(3)]
    assert [p.communicate(timeout=120)[0].strip() for p in procs] == ["ok 0", "ok 1", "ok 2"]
    assert not list(tmp_path.iterdir())


def test_bench_extras_guard_ends_a_hung_rank_with_the_headline():
    """the timer around the exchanging extra blocks of an N > 1 run: rank 0 prints the line it holds, and every
    rank ends NON-ZERO (a hang must not look like a green run: the launcher relays the line and fails)"""
    code = (
        "import sys, time, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "def emit(d):\n"
        "    print(json.dumps(d), flush=True)\n"
        "t = bench.start_extras_guard(emit, int(sys.argv[1]), {'value': 1.5, 'n_gpus': 2}, 1)\n"
        "time.sleep(30)\n"
        "print('not reached')\n")
    for rank, expect_line in ((0, True), (1, False)):
        out = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=20)
        assert out.returncode == 4 and "not reached" not in out.stdout and "timed out after 1 s" in out.stderr
        if expect_line:
            d = json.loads(out.stdout.strip())
            assert d["value"] == 1.5 and "did not finish within 1 s" in d["extras_error"]
        else:
            assert out.stdout.strip() == ""
    # cancelled in time: nothing happens
    code2 = code.replace("time.sleep(30)", "t.cancel(); time.sleep(2)").replace("print('not reached')", "print('done')")
    out = subprocess.run([sys.executable, "-c", code2, "0"], capture_output=True, text=True, timeout=20)
    assert out.returncode == 0 and out.stdout.strip() == "done"


def test_bench_spawn_environment():
    import bench

    base = {"PATH": "/usr/bin", "FC_BENCH_FORCE_SPAWN": "1"}
    envs = [bench.rank_env(base, r, 4, "/tmp/x.id") for r in range(4)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["FC_COMM_ID_FILE"] == "/tmp/x.id" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "FC_BENCH_FORCE_SPAWN" not in e and e["PATH"] == "/usr/bin" for e in envs)
    assert base == {"PATH": "/usr/bin", "FC_BENCH_FORCE_SPAWN": "1"}  # the parent's environment is untouched


def test_bench_ranks_never_import_torch():
    import re

    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"^\s*(import torch|from torch)", src, re.M) and "tdist" not in src
    # no exec of another program anywhere: ranks are child processes
    assert not re.search(r"os\.exec|execv", src)
