#!/bin/bash
set -u
OUT=gpurun_out/sweep_sched.txt; : > $OUT
for W in ${WORKLOADS:-cfg4 cfg5}; do for CL in 2 4 6 8; do
  echo "== $W cluster=$CL" >> $OUT
  DMRGX_CLUSTER=$CL timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 32 --warmup 8 >> $OUT 2>&1 || exit 1
done; done
