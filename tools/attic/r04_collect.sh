#!/bin/bash
# copies the summaries of gpurun_out/r04/ (tools/attic/r04_evidence.sh) into profiles/ under round-4 names
set -u
S=gpurun_out/r04; D=profiles
cp $S/bench_n1.json $D/r04_bench_n1.json
cp $S/bench_n1_k20.json $D/r04_bench_n1_k20.json
cp $S/bench_n1_under_rocprof.json $D/r04_bench_n1_under_rocprof.json
cp $S/bench_n1_kernel_stats.csv $D/r04_bench_n1_kernel_stats.csv
cp $S/pmc_complete/pmc_summary.txt $D/r04_pmc_complete.txt
cp $S/pmc_complete/kernel_stats.csv $D/r04_complete_kernel_stats.csv
cp $S/pmc_complete.json $D/r04_pmc_complete.json
cp $S/pmc_complete_a80/pmc_summary.txt $D/r04_pmc_complete_a80.txt
cp $S/pmc_complete_a80/kernel_stats.csv $D/r04_complete_a80_kernel_stats.csv
cp $S/pmc_complete_a80.json $D/r04_pmc_complete_a80.json
cp $S/complete_a80.json $D/r04_complete_a80_rates.jsonl
cp $S/pmc_refine/pmc_summary.txt $D/r04_pmc_refine.txt
cp $S/pmc_refine/kernel_stats.csv $D/r04_refine_kernel_stats.csv
cp $S/pmc_refine.json $D/r04_pmc_refine.json
for f in refine_alone.json refine_alone_round3_forms.json ladder_many.json ladder_one_workgroup.json cfg3_runs.json cfg3_kernel_stats.csv \
         cfg3_device_busy.json pin_probe.json hostin_breakdown.json hostin_breakdown_direct_uploads.json \
         bench_two_ranks_one_device_stub_collective.json gpu_tests.log workloads.jsonl bench_cfg4_n1.json bench_cfg5_n1.json; do
  [ -f $S/$f ] && cp $S/$f $D/r04_$f
done
ls -la $D | grep r04_ | wc -l
