#!/bin/bash
# Round-4 evidence.  Run on the GPU box from the repo root:  bash tools/attic/r04_evidence.sh [part ...]
# (parts: pmc pmc80 refine bench cfg3 tests workloads hostin ranks).  Writes gpurun_out/r04/; the summaries to keep are
# copied into profiles/ by tools/attic/r04_collect.sh on the authoring side.  Counter passes never share a run with tracing.
set -u
O=gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[r04] $*"; }
parts="${*:-pmc pmc80 refine bench cfg3 tests workloads hostin ranks}"
for part in $parts; do case $part in
pmc)
  say "PMC passes: complete alignment kernel, BASELINE configs[1] (tools/time_complete.py 10000 50 5)"
  bash tools/attic/r03_pmc.sh $O/pmc_complete r04 tools/time_complete.py 10000 50 5 > $O/pmc_complete.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_complete/pmc_summary.txt "k_simbits_screen_mfma<4, 2>" $O/pmc_complete.json stats=$O/pmc_complete/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="BASELINE configs[1]: 10000 x 50, fc_bench_rmsd_and_max_all"
  ;;
pmc80)
  say "PMC passes: complete alignment kernel at the cfg4 shape (35355 x 80: the <8, 2> variant)"
  bash tools/attic/r03_pmc.sh $O/pmc_complete_a80 r04a80 tools/time_complete.py 35355 80 2 > $O/pmc_complete_a80.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_complete_a80/pmc_summary.txt "k_simbits_screen_mfma<8, 2>" $O/pmc_complete_a80.json stats=$O/pmc_complete_a80/kernel_stats.csv n_conformers=35355 n_atoms=80 workload="cfg4 family N = 1 member: 35355 x 80, fc_bench_rmsd_and_max_all"
  python3 tools/time_complete.py 35355 80 3 > $O/complete_a80.json 2>/dev/null
  python3 tools/time_complete.py 12000 80 5 >> $O/complete_a80.json 2>/dev/null
  ;;
refine)
  say "PMC passes + stats: the long-queue refine on the continuous-RMSD ensemble (tools/refine_alone_probe.py)"
  bash tools/attic/r03_pmc.sh $O/pmc_refine r04ref tools/refine_alone_probe.py > $O/pmc_refine.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_refine/pmc_summary.txt "k_refine_buckets" $O/pmc_refine.json stats=$O/pmc_refine/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="continuous RMSD distribution, 945025 candidate pairs per launch"
  python3 tools/refine_alone_probe.py > $O/refine_alone.json 2>/dev/null
  FC_REFINE_BUCKETS=0 FC_LADDER_MANY=0 python3 tools/refine_alone_probe.py > $O/refine_alone_round3_forms.json 2>/dev/null
  python3 tools/attic/ladder_many_probe.py > $O/ladder_many.json 2>/dev/null
  FC_LADDER_MANY=0 python3 tools/attic/ladder_many_probe.py > $O/ladder_one_workgroup.json 2>/dev/null
  ;;
bench)
  say "bench default"; python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  say "bench driver form"; python bench.py --steps 20 --warmup 5 > $O/bench_n1_k20.json 2> $O/bench_n1_k20.err
  say "bench under rocprof (kernel trace + stats; the timed region only)"
  rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_under_rocprof.json 2> $O/prof_bench.err
  find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \;
  rm -rf $O/prof_bench
  ;;
cfg3)
  say "cfg3 csearch, 8 runs in one process; kernel trace of 4 runs"
  FC_CSEARCH_RUNS=8 python tools/bench_workloads.py csearch > $O/cfg3_runs.json 2> $O/cfg3_runs.err
  FC_CSEARCH_RUNS=4 rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 --output-format csv -- python3 tools/bench_workloads.py csearch > $O/cfg3_under_rocprof.json 2> $O/prof_cfg3.err
  find $O/prof_cfg3 -name "*kernel_stats.csv" -exec cp {} $O/cfg3_kernel_stats.csv \;
  find $O/prof_cfg3 -name "*kernel_trace.csv" -exec cp {} $O/cfg3_kernel_trace.csv \;
  rm -rf $O/prof_cfg3
  python3 tools/trace_busy.py $O/cfg3_kernel_trace.csv k_angle_grid > $O/cfg3_device_busy.json
  rm -f $O/cfg3_kernel_trace.csv
  ;;
tests)
  say "GPU tests (stdout AND stderr kept)"
  python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.log 2>&1; echo "exit code $?" >> $O/gpu_tests.log
  ;;
workloads)
  say "workloads plain"
  python tools/bench_workloads.py embed csearch prune80 cfg4 > $O/workloads.jsonl 2> $O/workloads.err
  python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
  python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
  ;;
hostin)
  say "host-in leg"
  python tools/attic/pin_probe.py > $O/pin_probe.json 2>/dev/null
  python tools/hostin_breakdown.py > $O/hostin_breakdown.json 2>/dev/null
  FC_STAGED_UPLOADS=0 python tools/hostin_breakdown.py > $O/hostin_breakdown_direct_uploads.json 2>/dev/null
  ;;
ranks)
  say "two ranks on one device with the stand-in collective (tests/stubs/rccl_stub.cpp): the N > 1 code path, not a scaling measurement"
  hipcc -O2 -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 tests/stubs/rccl_stub.cpp -o /tmp/librccl_stub.so -lrt -lpthread 2> $O/stub_build.err
  FC_BENCH_SAME_DEVICE=1 FC_RCCL_LIB=/tmp/librccl_stub.so python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_two_ranks_one_device_stub_collective.json 2> $O/bench_two_ranks_one_device_stub_collective.err
  ;;
esac; done
say done
