#!/bin/bash
# tools/ab_headline.sh <tag> <lib_a> <lib_b> [rounds] -- the headline step with two builds of the library on the SAME box,
# interleaved (boxes of the pool differ by 8 %): ms_per_step and the kernel's mean from the bench line, value_check kept
tag=$1; a=$2; b=$3; rounds=${4:-3}
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq 1 $rounds); do
  for lib in $a $b; do
    FC_LIB_PATH=$PWD/firecode_amd/$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/line.json 2> $out/err.log || { echo "bench failed ($lib)"; tail -5 $out/err.log; exit 1; }
    python - "$lib" $out/line.json <<'PY' | tee -a $out/ab.log
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], "check", d.get("value_check", {}).get("ok"))
PY
  done
done
