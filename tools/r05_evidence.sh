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
benchprof)
  say "bench under rocprofv3 (kernel trace + stats; the timed region only)"
  cd /tmp && cd "$GRAFT_REPO_ROOT"
  rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_under_rocprof.json 2> $O/prof_bench.err || exit 1
  find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \;
  rm -rf $O/prof_bench
  head -4 $O/bench_n1_kernel_stats.csv | cut -c1-200
  ;;
pmc)
  say "PMC passes: complete alignment kernel, BASELINE configs[1] (tools/time_complete.py 10000 50 5)"
  bash tools/attic/r03_pmc.sh $O/pmc_complete r05 tools/time_complete.py 10000 50 5 > $O/pmc_complete.log 2>&1 || exit 1
  python3 tools/attic/r03_pmc_json.py $O/pmc_complete/pmc_summary.txt "k_simbits_screen_mfma<4, 2, 64, true>" $O/pmc_complete.json stats=$O/pmc_complete/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="BASELINE configs[1]: 10000 x 50, fc_bench_rmsd_and_max_all"
  cp $O/pmc_complete/pmc_summary.txt $O/pmc_complete.txt
  rm -rf $O/pmc_complete/pmc_* $O/pmc_complete/trace
  cut -c1-400 $O/pmc_complete.json
  ;;
hostin)
  say "host arrays in -> mask out: breakdown (four runs), device timeline of one call, the same from a page-locked source"
  : > $O/hostin_breakdown.json
  for i in 1 2 3 4; do python tools/hostin_breakdown.py >> $O/hostin_breakdown.json 2>>$O/hostin.err || exit 1; done
  cd /tmp && cd "$GRAFT_REPO_ROOT"
  rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_hostin --output-format csv -- python3 tools/hostin_timeline.py > $O/hostin_calls.json 2>>$O/hostin.err || exit 1
  python3 tools/hostin_timeline.py $O/prof_hostin > $O/hostin_timeline.txt
  rm -rf $O/prof_hostin
  FC_TIMELINE_PINNED=1 python3 tools/hostin_timeline.py > $O/hostin_calls_pinned.json 2>>$O/hostin.err
  cat $O/hostin_breakdown.json $O/hostin_calls.json $O/hostin_calls_pinned.json; cat $O/hostin_timeline.txt
  ;;
pmc80)
  say "PMC passes: complete alignment kernel at the cfg4 shape (35355 x 80: the <8, 2, 64, true> variant)"
  bash tools/attic/r03_pmc.sh $O/pmc_complete_a80 r05a80 tools/time_complete.py 35355 80 2 > $O/pmc_complete_a80.log 2>&1 || exit 1
  python3 tools/attic/r03_pmc_json.py $O/pmc_complete_a80/pmc_summary.txt "k_simbits_screen_mfma<8, 2, 64, true>" $O/pmc_complete_a80.json stats=$O/pmc_complete_a80/kernel_stats.csv n_conformers=35355 n_atoms=80 workload="cfg4 family N = 1 member: 35355 x 80, fc_bench_rmsd_and_max_all"
  cp $O/pmc_complete_a80/pmc_summary.txt $O/pmc_complete_a80.txt
  rm -rf $O/pmc_complete_a80/pmc_* $O/pmc_complete_a80/trace
  python3 tools/time_complete.py 35355 80 3 > $O/complete_a80.json 2>/dev/null
  cat $O/complete_a80.json
  ;;
screenpmc)
  say "PMC passes + stats: the split-half screen on the prune path, cfg4 N = 1 member (35355 x 80) and BASELINE configs[1]"
  bash tools/attic/r03_pmc.sh $O/pmc_cfg4 r05cfg4 tools/attic/prune_probe.py 35355 80 6 20 > $O/pmc_cfg4.log 2>&1 || exit 1
  python3 tools/attic/r03_pmc_json.py $O/pmc_cfg4/pmc_summary.txt "k_simbits_screen_mfma_h2" $O/pmc_screen_h2_cfg4_member.json stats=$O/pmc_cfg4/kernel_stats.csv n_conformers=35355 n_atoms=80 workload="cfg4 family, n_gpus = 1 member"
  cp $O/pmc_cfg4/pmc_summary.txt $O/pmc_cfg4_member.txt; cp $O/pmc_cfg4/kernel_stats.csv $O/cfg4_member_kernel_stats.csv
  rm -rf $O/pmc_cfg4/pmc_* $O/pmc_cfg4/trace
  bash tools/attic/r03_pmc.sh $O/pmc_prune r05prune tools/attic/prune_probe.py 10000 50 2 100 > $O/pmc_prune.log 2>&1 || exit 1
  python3 tools/attic/r03_pmc_json.py $O/pmc_prune/pmc_summary.txt "k_simbits_screen_mfma_h2" $O/pmc_screen_h2.json stats=$O/pmc_prune/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="BASELINE configs[1], prune path"
  cp $O/pmc_prune/pmc_summary.txt $O/pmc_prune.txt; cp $O/pmc_prune/kernel_stats.csv $O/prune_kernel_stats.csv
  rm -rf $O/pmc_prune/pmc_* $O/pmc_prune/trace
  cut -c1-300 $O/pmc_screen_h2_cfg4_member.json; cut -c1-300 $O/pmc_screen_h2.json
  ;;
refine)
  say "PMC passes + stats: the long-queue refine on the continuous-RMSD ensemble (tools/refine_alone_probe.py)"
  bash tools/attic/r03_pmc.sh $O/pmc_refine r05ref tools/refine_alone_probe.py > $O/pmc_refine.log 2>&1 || exit 1
  python3 tools/attic/r03_pmc_json.py $O/pmc_refine/pmc_summary.txt "k_refine_buckets" $O/pmc_refine.json stats=$O/pmc_refine/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="continuous RMSD distribution, 945025 candidate pairs per launch"
  cp $O/pmc_refine/pmc_summary.txt $O/pmc_refine.txt
  rm -rf $O/pmc_refine/pmc_* $O/pmc_refine/trace
  python3 tools/refine_alone_probe.py > $O/refine_alone.json 2>/dev/null
  cut -c1-300 $O/pmc_refine.json; cat $O/refine_alone.json
  ;;
*) echo "unknown part $1"; exit 2 ;;
esac
