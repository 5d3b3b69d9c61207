# tools/attic/r03_pmc.sh OUTDIR TAG SCRIPT [ARGS...] -- rocprofv3 counter passes (one --pmc set per run, never mixed
# with tracing) + one --kernel-trace --stats run of `python3 SCRIPT ARGS`, summaries into OUTDIR.
set -u
O=$1; TAG=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  echo "[pmc] $tag"
  rocprofv3 --pmc $pass -d $O/pmc_$tag --output-format csv -- python3 "$@" > $O/pmc_$tag.log 2>&1
done
python3 tools/pmc_summary.py $TAG $O/pmc_* > $O/pmc_summary.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 "$@" > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep -v "rocclr\|^$" $O/pmc_summary.txt | cut -c1-1500
head -8 $O/kernel_stats.csv
