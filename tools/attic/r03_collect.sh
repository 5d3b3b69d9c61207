#!/bin/bash
# copies the summaries of gpurun_out/r03/ (tools/attic/r03_evidence.sh) into profiles/ under round-3 names
set -u
S=gpurun_out/r03; D=profiles
cp $S/bench_n1.json $D/r03_bench_n1.json
cp $S/bench_n1_k20.json $D/r03_bench_n1_k20.json
cp $S/bench_n1_under_rocprof.json $D/r03_bench_n1_under_rocprof.json
cp $S/bench_n1_kernel_stats.csv $D/r03_bench_n1_kernel_stats.csv
cp $S/pmc_complete/pmc_summary.txt $D/r03_pmc_complete.txt
cp $S/pmc_complete/kernel_stats.csv $D/r03_complete_kernel_stats.csv
cp $S/pmc_complete.json $D/r03_pmc_complete.json
cp $S/pmc_secondary/pmc_summary.txt $D/r03_pmc_secondary.txt
cp $S/pmc_secondary/kernel_stats.csv $D/r03_secondary_kernel_stats.csv
cp $S/pmc_refine.json $D/r03_pmc_refine.json
cp $S/secondary_overlapped.json $D/r03_secondary_overlapped.json
cp $S/pmc_cfg4/pmc_summary.txt $D/r03_pmc_cfg4_member.txt
cp $S/pmc_cfg4/kernel_stats.csv $D/r03_cfg4_member_kernel_stats.csv
cp $S/pmc_screen_h2_cfg4.json $D/r03_pmc_screen_h2_cfg4_member.json
cp $S/pmc_prune/pmc_summary.txt $D/r03_pmc_prune.txt
cp $S/pmc_prune/kernel_stats.csv $D/r03_prune_kernel_stats.csv
cp $S/pmc_screen_h2.json $D/r03_pmc_screen_h2.json
cp $S/workloads_under_rocprof.jsonl $D/r03_workloads_under_rocprof.jsonl
cp $S/workloads_kernel_stats.csv $D/r03_workloads_kernel_stats.csv
cp $S/workloads.jsonl $D/r03_workloads.jsonl
cp $S/bench_spawned_1rank.json $D/r03_bench_spawned_1rank.json
cp $S/bench_forced_sharded_1rank.json $D/r03_bench_forced_sharded_1rank.json
cp $S/bench_cfg4_n1.json $D/r03_bench_cfg4_family_n1.json
cp $S/bench_cfg5_n1.json $D/r03_bench_cfg5_family_n1.json
for f in cfg3_runs.json cfg3_runs_host_grid.json issue_model_f16mfma_valu.txt issue_model_f16mfma_valu_vop2.txt issue_model_f16mfma_valu_pk.txt two_ranks_one_device.txt two_ranks_one_device_rank0.json prune_step_timeline.txt torchrun_2ranks_one_device.json; do
  [ -f $S/$f ] && cp $S/$f $D/r03_$f
done
ls -la $D | grep r03_ | wc -l
