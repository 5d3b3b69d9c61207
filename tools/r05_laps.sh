#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
FC_DEBUG=1 FC_SCAN_LAPS=1 FC_CSEARCH_RUNS=4 timeout -k 10 300 python tools/bench_workloads.py csearch > $O/laps.json 2> $O/laps.err
grep -E "\[fc\]" $O/laps.err | tail -40
