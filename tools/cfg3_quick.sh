for lib in libfc_hip.so libfc_hip_ts4.so; do
FC_LIB_PATH=firecode_amd/$lib FC_CSEARCH_RUNS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/q_$lib --output-format csv -- python3 tools/bench_workloads.py csearch > gpurun_out/q_$lib.json 2>/dev/null
python - $lib <<'PY'
import csv,glob,sys,json
f=glob.glob("gpurun_out/q_%s/**/*kernel_stats.csv"%sys.argv[1],recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_ts_level" in r["Name"]: print(sys.argv[1],"k_ts_level", r["Calls"], r["TotalDurationNs"], r["MaxNs"])
d=json.load(open("gpurun_out/q_%s.json"%sys.argv[1])); print([round(r["s_total"]*1e3,2) for r in d["all_runs"]])
PY
done
