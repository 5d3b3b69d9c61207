#!/bin/bash
# round 5: the device ladder on the GPU box -- tests first (stderr kept), then cfg3 timings
set -o pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 900 python -m pytest tests/test_tfd_gpu_graph.py -x -q > gpurun_out/r5a/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5a/tests.log
[ $rc -ne 0 ] && exit $rc
FC_DEBUG=1 timeout -k 10 300 python tools/dump_cfg3_fm.py > gpurun_out/r5a/ladder_debug.log 2>&1 || exit 1
grep -E "ladder_call_s|tfd ladder \(device\)" gpurun_out/r5a/ladder_debug.log | tail -8
timeout -k 10 300 python tools/bench_workloads.py csearch > gpurun_out/r5a/cfg3.json 2>gpurun_out/r5a/cfg3.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/cfg3.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("after_tfd", "after_rmsd", "s_total")}, d["second_run"])
PY
