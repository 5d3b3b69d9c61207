#!/bin/bash
# tools/r05_evidence.sh <part> -- round 5 evidence on the GPU box (outputs under gpurun_out/r05; tools/r05_collect.sh copies what is kept)
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
say() { echo "== $*"; }
case "$1" in
cfg3)
  say "cfg3 csearch, 8 runs in one process; kernel trace of 4 runs"
  FC_CSEARCH_RUNS=8 python tools/bench_workloads.py csearch > $O/cfg3_runs.json 2> $O/cfg3_runs.err || exit 1
  FC_CSEARCH_RUNS=4 rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 --output-format csv -- python3 tools/bench_workloads.py csearch > $O/cfg3_under_rocprof.json 2> $O/prof_cfg3.err || exit 1
  find $O/prof_cfg3 -name "*kernel_stats.csv" -exec cp {} $O/cfg3_kernel_stats.csv \;
  find $O/prof_cfg3 -name "*kernel_trace.csv" -exec cp {} $O/cfg3_kernel_trace.csv \;
  rm -rf $O/prof_cfg3
  python3 tools/trace_busy.py $O/cfg3_kernel_trace.csv k_angle_grid > $O/cfg3_device_busy.json
  python3 tools/trace_phases.py $O/cfg3_kernel_trace.csv k_angle_grid > $O/cfg3_phases.txt
  rm -f $O/cfg3_kernel_trace.csv
  cat $O/cfg3_device_busy.json | head -12
  ;;
ladder)
  say "the TFD ladder alone on the cfg3 first-match array, kernel trace"
  rocprofv3 --kernel-trace --stats -d $O/prof_ladder --output-format csv -- python3 tools/ladder_probe.py 4 > $O/ladder_probe.json 2> $O/ladder_probe.err || exit 1
  find $O/prof_ladder -name "*kernel_stats.csv" -exec cp {} $O/ladder_kernel_stats.csv \;
  mkdir -p $O/ladder/prof/x && find $O/prof_ladder -name "*kernel_trace.csv" -exec cp {} $O/ladder/prof/x/1_kernel_trace.csv \;
  python3 tools/ladder_timeline.py $O/ladder > $O/ladder_timeline.txt
  rm -rf $O/prof_ladder $O/ladder
  cat $O/ladder_probe.json
  ;;
tests)
  say "GPU tests (stdout AND stderr kept)"
  timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.log 2>&1; echo "exit code $?" >> $O/gpu_tests.log
  tail -5 $O/gpu_tests.log
  ;;
bench)
  say "bench as the driver runs it"
  python bench.py --steps 20 --warmup 5 > $O/bench_n1_k20.json 2> $O/bench_n1_k20.err
  tail -c 600 $O/bench_n1_k20.json
  ;;
atoms)
  say "complete alignments over the atom count (8 000 conformers): 64-, 32- and 16-column tiles, the exact kernel beyond"
  : > $O/complete_over_atoms.jsonl
  for a in 24 50 52 53 80 104 105 128 160 200 208 209 260 320 416 420; do
    timeout -k 10 200 python tools/time_complete.py 8000 $a 3 2>/dev/null | tail -1 >> $O/complete_over_atoms.jsonl || exit 1
  done
  cat $O/complete_over_atoms.jsonl | cut -c1-160
  ;;
laps)
  say "cfg3 search with per-phase laps on stderr (FC_DEBUG FC_SCAN_LAPS)"
  FC_DEBUG=1 FC_SCAN_LAPS=1 FC_CSEARCH_RUNS=4 timeout -k 10 300 python tools/bench_workloads.py csearch > $O/cfg3_laps.json 2> $O/cfg3_laps.err
  grep -E "\[fc\]" $O/cfg3_laps.err | tail -24
  ;;
*) echo "unknown part $1"; exit 2 ;;
esac
