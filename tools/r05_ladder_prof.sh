#!/bin/bash
# tools/r05_ladder_prof.sh <tag>: TFD ladder tests (stderr kept), then the ladder alone under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_tfd_gpu_graph.py -x -q > $O/tests.log 2>&1
rc=$?
tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 tools/ladder_probe.py 4 > $O/probe.json 2> $O/probe.err || exit 1
cat $O/probe.json
