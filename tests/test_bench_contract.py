"""bench.py prints ONE JSON line with the contract's fields -- on the single-GPU path and on the
multi-GPU code path (RCCL communicator of one rank through the C ABI), whose stdout RCCL would
otherwise share -- and never imports torch."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(extra_env, args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("sharded", [False, True])
def test_bench_prints_one_contract_line(sharded):
    d = _run({"FC_BENCH_FORCE_SHARDED": "1"} if sharded else {}, ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == "conformer-pair RMSD alignments/s" and d["unit"] == "alignments/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    # the arithmetic of the dominant kernel decides the label; the decisions themselves are fp64 either way
    assert (d["dtype"], d["roofline"]["dtype"], d["roofline"]["peak"]) in (
        ("f16x2 screen (split-half, fp32-accurate) + f64 exact refine", "f16x2", 2500.0),
        ("f32 screen + f64 exact refine", "f32", 157.3), ("f64", "f64", 78.6))
    assert "workload" in d["config"] and d["survivor_count_ok"] is True and d["survivors_are_last_cluster_members"] is True
    assert d["config"]["baseline_config"] == "configs[1]" and "pair decisions" in d["config"]["value_counts"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(d["value"] - d["config"]["pairs_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    if sharded:
        assert d["config"]["comm"].startswith("RCCL through libfc_hip.so")
    else:
        # the stricter readings ride in the same line, each with its own device timing and roofline
        f = d["fp64_path"]
        assert f["dtype"] == "f64" and f["roofline"]["peak"] == 78.6 and 0 < f["roofline"]["frac"] < 1
        assert f["mask_equals_default_path"] is True and f["value"] > 0
        assert d["alignments_complete_per_s"] > 0 and d["alignments_complete"]["dtype"] == "f64"
        assert d["rmsd_values_per_s"] > 0 and d["config"]["secondary"]["ms_per_step"] > 0
        assert d["pruned_ensembles_per_s_host_in_mask_out"] > 0
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic_source"].startswith("from_file")


@pytest.mark.gpu
def test_bench_cpu_baseline_block():
    d = _run({"FC_BENCH_CPU_SECONDS": "3"}, ["--steps", "1", "--warmup", "0"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "alignments/s" and c["value"] > 0 and c["sample"]


@pytest.mark.gpu
def test_bench_cfg4_and_cfg5_workloads_run_on_one_rank():
    d = _run({}, ["--workload", "cfg4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["config"]["baseline_config"] == "configs[3]" and d["config"]["n_atoms"] == 80
    assert d["survivor_count_ok"] is True and d["survivors_are_last_cluster_members"] is True and d["scaling"] == "weak"
    d = _run({}, ["--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert d["config"]["baseline_config"] == "configs[4]" and d["unit"] == "poses/s" and d["value"] > 0


def test_bench_ranks_never_import_torch():
    import re

    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"^\s*(import torch|from torch)", src, re.M) and "tdist" not in src
