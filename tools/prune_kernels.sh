#!/bin/bash
# tools/prune_kernels.sh <variant ...> -- per-kernel mean durations of the synchronous prune of the continuous-RMSD
# ensemble (tools/attic/ladder_many_probe.py under rocprofv3 --kernel-trace --stats) for builds of the library:
# `base` = firecode_amd/libfc_hip.so, `x` = firecode_amd/libfc_hip_x.so (make BUILD=build_x OUT=../libfc_hip_x.so EXTRA=-D...).
# Ablation macros of fc_kabsch.hip (timing only, results wrong): FC_ABLATE_STAGE, FC_ABLATE_OVERFLOW, FC_ABLATE_REDO.
mkdir -p gpurun_out/prune_kernels && cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  [ "$v" = base ] && lib=libfc_hip.so || lib=libfc_hip_$v.so
  FC_LIB_PATH=$PWD/firecode_amd/$lib rocprofv3 --kernel-trace --stats -d gpurun_out/prune_kernels/p_$v --output-format csv -- python3 tools/attic/ladder_many_probe.py > gpurun_out/prune_kernels/o_$v.json 2> gpurun_out/prune_kernels/e_$v.err
  echo "variant [$v]"; cat gpurun_out/prune_kernels/o_$v.json
  f=$(find gpurun_out/prune_kernels/p_$v -name "*kernel_stats.csv"); python3 tools/kstats.py $f screen_mfma k_refine k_ladder k_pair_buckets k_bucket k_screen_verdict
  rm -rf gpurun_out/prune_kernels/p_$v
done
