#!/bin/bash
# Round-3 evidence.  Run on the GPU box from the repo root:  bash tools/attic/r03_evidence.sh [part ...]
# (parts: pmc secondary cfg4 bench workloads variants extra; default all -- the counter parts first: bench.py quotes the
# kept counter files, so collect (tools/attic/r03_collect.sh) after them and run `bench` again for lines that quote this run's files).  Writes gpurun_out/r03/; the summaries to keep
# are copied into profiles/ by tools/attic/r03_collect.sh on the authoring side.
set -u
O=gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[r03] $*"; }
parts="${*:-pmc secondary cfg4 bench workloads variants extra}"
for part in $parts; do case $part in
bench)
  say "bench default"; python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  say "bench driver form"; python bench.py --steps 20 --warmup 5 > $O/bench_n1_k20.json 2> $O/bench_n1_k20.err
  say "bench under rocprof (kernel trace + stats; the timed region only)"
  rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_under_rocprof.json 2> $O/prof_bench.err
  find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \;
  ;;
pmc)
  say "PMC passes: complete alignment kernel (tools/time_complete.py 10000 50 5)"
  bash tools/attic/r03_pmc.sh $O/pmc_complete r03 tools/time_complete.py 10000 50 5 > $O/pmc_complete.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_complete/pmc_summary.txt "k_simbits_screen_mfma<4, 2>" $O/pmc_complete.json stats=$O/pmc_complete/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="BASELINE configs[1]: 10000 x 50, fc_bench_rmsd_and_max_all"
  ;;
secondary)
  say "PMC passes + stats: continuous-RMSD ensemble (tools/attic/secondary_probe.py), kernels one after another"
  FC_BENCH_LANES=1 bash tools/attic/r03_pmc.sh $O/pmc_secondary r03sec tools/attic/secondary_probe.py 10 > $O/pmc_secondary.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_secondary/pmc_summary.txt "k_refine_pairs" $O/pmc_refine.json stats=$O/pmc_secondary/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="continuous RMSD distribution, 894764 candidate pairs per launch"
  python3 tools/attic/secondary_probe.py 20 > $O/secondary_overlapped.json 2>/dev/null
  ;;
cfg4)
  say "PMC passes + stats: cfg4-family N = 1 member (35355 x 80), prune path"
  bash tools/attic/r03_pmc.sh $O/pmc_cfg4 r03cfg4 tools/attic/prune_probe.py 35355 80 6 20 > $O/pmc_cfg4.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_cfg4/pmc_summary.txt "k_simbits_screen_mfma_h2" $O/pmc_screen_h2_cfg4.json stats=$O/pmc_cfg4/kernel_stats.csv n_conformers=35355 n_atoms=80 workload="cfg4 family, n_gpus = 1 member"
  say "PMC passes: prune path at cfg2"
  bash tools/attic/r03_pmc.sh $O/pmc_prune r03prune tools/attic/prune_probe.py 10000 50 2 100 > $O/pmc_prune.log 2>&1
  python3 tools/attic/r03_pmc_json.py $O/pmc_prune/pmc_summary.txt "k_simbits_screen_mfma_h2" $O/pmc_screen_h2.json stats=$O/pmc_prune/kernel_stats.csv n_conformers=10000 n_atoms=50 workload="BASELINE configs[1], prune path"
  ;;
workloads)
  say "workloads under rocprof"
  rocprofv3 --kernel-trace --stats -d $O/prof_workloads --output-format csv -- python3 tools/bench_workloads.py embed csearch prune80 tri values > $O/workloads_under_rocprof.jsonl 2> $O/prof_workloads.err
  find $O/prof_workloads -name "*kernel_stats.csv" -exec cp {} $O/workloads_kernel_stats.csv \;
  say "workloads plain"
  python tools/bench_workloads.py embed csearch prune80 cfg4 pcie queue > $O/workloads.jsonl 2> $O/workloads.err
  ;;
variants)
  say "bench: self-spawn with one rank, forced sharded, cfg4 line, cfg5 line"
  FC_BENCH_FORCE_SPAWN=1 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_spawned_1rank.json 2> $O/bench_spawned_1rank.err
  FC_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_forced_sharded_1rank.json 2> $O/bench_forced_sharded_1rank.err
  python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
  python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
  ;;
extra)
  say "cfg3 csearch, 8 runs in one process (first / second / steady)"
  FC_CSEARCH_RUNS=8 python tools/bench_workloads.py csearch > $O/cfg3_runs.json 2> $O/cfg3_runs.err
  FC_CSEARCH_HOST_GRID=1 FC_CSEARCH_RUNS=4 python tools/bench_workloads.py csearch > $O/cfg3_runs_host_grid.json 2>> $O/cfg3_runs.err
  say "issue model of one SIMD beside f16 MFMAs (three encodings of the vector instruction)"
  ./tools/ubench_issue_model > $O/issue_model_f16mfma_valu.txt 2>&1
  ./tools/ubench_issue_model_vop2 > $O/issue_model_f16mfma_valu_vop2.txt 2>&1
  ./tools/ubench_issue_model_pk > $O/issue_model_f16mfma_valu_pk.txt 2>&1
  say "two bench ranks on one device (id file rendezvous, ncclCommInitRank, file fallback)"
  timeout -k 10 400 bash tools/attic/two_ranks_one_device.sh --no-extras > $O/two_ranks_one_device.txt 2>&1
  cp gpurun_out/two_r0.out $O/two_ranks_one_device_rank0.json 2>/dev/null
  say "the driver's launch form, two ranks on one device"
  FC_BENCH_SAME_DEVICE=1 FC_COMM_TIMEOUT_S=120 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/torchrun_2ranks_one_device.json 2> $O/torchrun_2ranks_one_device.err
  say "timeline of steady-state prune steps"
  rocprofv3 --kernel-trace -d $O/prof_timeline --output-format csv -- python3 tools/attic/prune_probe.py 10000 50 2 100 > $O/timeline_probe.json 2> $O/timeline.err
  python3 tools/step_timeline.py $(find $O/prof_timeline -name "*kernel_trace.csv" | head -1) 60 3 > $O/prune_step_timeline.txt 2>&1
  ;;
esac; done
say "done"; ls $O | head -80
