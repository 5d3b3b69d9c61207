# first-match step of cfg3 alone: correctness + kernel stats of the 16-bit path over its knob, phase stamps of the walk
set -e
mkdir -p gpurun_out/fm
timeout -k 10 200 python tools/fm_probe.py
for ahead in 32 64 128 256; do
  FC_TFD_LOOKAHEAD=$ahead rocprofv3 --kernel-trace --stats -d gpurun_out/fm/prof_$ahead --output-format csv -- python3 tools/fm_probe.py > /dev/null 2>&1
  python - $ahead <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/fm/prof_%s/**/*kernel_stats.csv"%(sys.argv[1]),recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    if "first_match" in r["Name"]: out.append((r["Name"].split("(")[0][-28:], round(float(r["AverageNs"])/1e3)))
print("ahead",sys.argv[1],out, flush=True)
PY
done
FC_LIB_PATH=firecode_amd/libfc_hip_stamps.so timeout -k 10 200 python tools/fm_stamps.py
