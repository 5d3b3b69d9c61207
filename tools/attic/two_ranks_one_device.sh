#!/bin/bash
# two bench ranks on ONE device: RCCL refuses (or accepts) the duplicate device; either way one JSON line from rank 0
ID=/tmp/fc_two_$$.id
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 FC_COMM_ID_FILE=$ID FC_COMM_TIMEOUT_S=60 \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/two_r$r.out 2> gpurun_out/two_r$r.err &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=$?; done
echo "exit $rc"
tail -c 1500 gpurun_out/two_r0.out; echo; tail -5 gpurun_out/two_r0.err; tail -5 gpurun_out/two_r1.err
