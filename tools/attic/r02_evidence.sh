#!/bin/bash
# Round-2 evidence: bench lines, rocprofv3 kernel stats and PMC passes.  Run on the GPU box from the repo root:
#   bash tools/attic/r02_evidence.sh      (writes gpurun_out/r02ev/; the summaries to keep are copied into profiles/ by hand)
set -u
O=gpurun_out/r02ev
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[r02ev] $*"; }
say "bench default"; python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
say "bench driver form"; python bench.py --steps 20 --warmup 5 > $O/bench_n1_k20.json 2> $O/bench_n1_k20.err
say "bench under rocprof (kernel trace + stats)"
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_under_rocprof.json 2> $O/prof_bench.err
say "bench fp64 screen under rocprof"
FC_SCREEN_F32=0 rocprofv3 --kernel-trace --stats -d $O/prof_bench_f64 --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_f64_under_rocprof.json 2> $O/prof_bench_f64.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  say "pmc pass $tag (f32 screen)"
  rocprofv3 --pmc $pass -d $O/pmc_f32_$tag --output-format csv -- python3 tools/screen_probe.py > $O/pmc_f32_$tag.log 2>&1
done
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  say "pmc pass $tag (f64 screen)"
  FC_SCREEN_F32=0 rocprofv3 --pmc $pass -d $O/pmc_f64_$tag --output-format csv -- python3 tools/screen_probe.py > $O/pmc_f64_$tag.log 2>&1
done
python tools/pmc_summary.py r02f32 $O/pmc_f32_* > $O/pmc_f32_summary.txt 2>&1
python tools/pmc_summary.py r02f64 $O/pmc_f64_* > $O/pmc_f64_summary.txt 2>&1
say "workloads under rocprof"
rocprofv3 --kernel-trace --stats -d $O/prof_workloads --output-format csv -- python3 tools/bench_workloads.py embed csearch prune80 tri values > $O/workloads.jsonl 2> $O/prof_workloads.err
say "workloads plain"
python tools/bench_workloads.py embed csearch prune80 cfg4 pcie queue > $O/workloads_plain.jsonl 2> $O/workloads_plain.err
python tools/broad_probe.py >> $O/workloads_plain.jsonl 2>> $O/workloads_plain.err
say "bench cfg4 / cfg5 / forced sharded on one rank"
python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
FC_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_forced_sharded_1rank.json 2> $O/bench_forced_sharded_1rank.err
say "done"; ls $O | head -50
