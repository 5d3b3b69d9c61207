# k_ts_level (sum of 32 launches, slowest launch, ns) and the search times of library builds side by side:
#   bash tools/cfg3_quick.sh [suffix ...]     (firecode_amd/libfc_hip_<suffix>.so; the default library first)
for lib in "" "$@"; do
p=libfc_hip${lib:+_$lib}.so
FC_LIB_PATH=firecode_amd/$p FC_CSEARCH_RUNS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/q_$p --output-format csv -- python3 tools/bench_workloads.py csearch > gpurun_out/q_$p.json 2>/dev/null
python - $p <<'PY'
import csv,glob,sys,json
f=glob.glob("gpurun_out/q_%s/**/*kernel_stats.csv"%sys.argv[1],recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_ts_level" in r["Name"]: print(sys.argv[1],"k_ts_level", r["Calls"], r["TotalDurationNs"], r["MaxNs"])
d=json.load(open("gpurun_out/q_%s.json"%sys.argv[1])); print([round(r["s_total"]*1e3,2) for r in d["all_runs"]])
PY
done
