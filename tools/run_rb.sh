# the long-queue refine alone over tuning builds (firecode_amd/libfc_hip_rb*.so): refine_ms of tools/refine_alone_probe.py
set -e
echo "default"; timeout -k 10 200 python tools/refine_alone_probe.py
for v in "$@"; do
  lib=${v%%:*}; grid=${v##*:}
  echo "$v"
  if [ "$grid" != "$lib" ]; then FC_REFINE_GRID=$grid FC_LIB_PATH=firecode_amd/libfc_hip_$lib.so timeout -k 10 200 python tools/refine_alone_probe.py
  else FC_LIB_PATH=firecode_amd/libfc_hip_$lib.so timeout -k 10 200 python tools/refine_alone_probe.py; fi
done
