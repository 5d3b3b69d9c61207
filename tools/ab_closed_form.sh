for k in 1 0 1 0; do
FC_SCAN_CLOSED_FORM=$k FC_CSEARCH_RUNS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$k --output-format csv -- python3 tools/bench_workloads.py csearch > gpurun_out/ab_$k.json 2>/dev/null
python - $k <<'PY'
import csv,glob,sys,json
f=sorted(glob.glob("gpurun_out/ab_%s/**/*kernel_stats.csv"%sys.argv[1],recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "k_ts_level" in r["Name"]: print("closed form", sys.argv[1], "k_ts_level total ns", r["TotalDurationNs"], "max", r["MaxNs"])
PY
rm -rf gpurun_out/ab_$k
done
