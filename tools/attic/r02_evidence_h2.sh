#!/bin/bash
# Round-2 evidence after the split-half screen (kind 16) became the default.  Run on the GPU box from the repo root:
#   bash tools/attic/r02_evidence_h2.sh      (writes gpurun_out/r02h2/; the summaries to keep are copied into profiles/ by hand)
set -u
O=gpurun_out/r02h2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[r02h2] $*"; }
say "bench default"; python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
say "bench driver form"; python bench.py --steps 20 --warmup 5 > $O/bench_n1_k20.json 2> $O/bench_n1_k20.err
say "bench under rocprof (kernel trace + stats)"
rocprofv3 --kernel-trace --stats -d $O/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_n1_under_rocprof.json 2> $O/prof_bench.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  say "pmc pass $tag (split-half screen)"
  rocprofv3 --pmc $pass -d $O/pmc_h2_$tag --output-format csv -- python3 tools/screen_probe.py > $O/pmc_h2_$tag.log 2>&1
done
python tools/pmc_summary.py r02h2 $O/pmc_h2_* > $O/pmc_h2_summary.txt 2>&1
say "model check, accumulators against fp64, A/B of the three matrix-pipe screens"
python tools/h2_probe.py > $O/h2_probe.jsonl 2> $O/h2_probe.err
./tools/ubench_mfma_f16_numerics > $O/mfma_f16_numerics.txt 2>&1
say "workloads under rocprof"
rocprofv3 --kernel-trace --stats -d $O/prof_workloads --output-format csv -- python3 tools/bench_workloads.py embed csearch prune80 tri values > $O/workloads_under_rocprof.jsonl 2> $O/prof_workloads.err
say "workloads plain"
python tools/bench_workloads.py embed csearch prune80 cfg4 pcie queue > $O/workloads.jsonl 2> $O/workloads.err
python tools/broad_probe.py >> $O/workloads.jsonl 2>> $O/workloads.err
say "bench cfg4 / cfg5 / forced sharded on one rank"
python bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
FC_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_forced_sharded_1rank.json 2> $O/bench_forced_sharded_1rank.err
say "done"; ls $O | head -60
