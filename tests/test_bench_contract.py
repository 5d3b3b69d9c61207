"""bench.py prints ONE JSON line with the contract's fields -- on the single-GPU path and on the
multi-GPU code path (RCCL group of one rank), whose stdout RCCL would otherwise share."""

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
        ("f32 screen + f64 exact refine", "f32", 157.3), ("f64", "f64", 78.6))
    assert "workload" in d["config"] and d["mask_ok"] is True
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(d["value"] - d["config"]["pairs_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    if sharded:
        assert d["config"]["exchange"].startswith("device-resident")


@pytest.mark.gpu
def test_bench_cpu_baseline_block():
    d = _run({"FC_BENCH_CPU_SECONDS": "3"}, ["--steps", "1", "--warmup", "0"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "alignments/s" and c["value"] > 0 and c["sample"]
