#!/bin/bash
# tools/r05_cfg3.sh <tag>: ladder tests, ladder profile, the cfg3 search (timings + kernel trace)
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_tfd_gpu_graph.py -x -q > $O/tests.log 2>&1
rc=$?
tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 tools/ladder_probe.py 4 > $O/probe.json 2> $O/probe.err || exit 1
cat $O/probe.json
FC_CSEARCH_RUNS=7 timeout -k 10 300 python tools/bench_workloads.py csearch > $O/cfg3.json 2> $O/cfg3.err || exit 1
python3 - <<PY
import json
d = json.loads(open("$O/cfg3.json").read().strip().splitlines()[-1])
print("cfg3", d["after_tfd"], d["after_rmsd"], "s_total", d["s_total"], "all", d.get("all_runs"))
PY
