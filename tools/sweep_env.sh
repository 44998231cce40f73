#!/bin/bash
# A/B of an environment toggle on the same box: usage VAR=DMRGX_SCHED_TOTAL tools/sweep_env.sh
set -u
OUT=gpurun_out/sweep_sched.txt; : > $OUT
for W in ${WORKLOADS:-cfg4 cfg5}; do for rep in 1 2; do for v in off on; do
  echo "== $W ${VAR}=$v rep$rep" >> $OUT
  if [ $v = on ]; then export ${VAR}=1; else unset ${VAR}; fi
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 32 --warmup 8 >> $OUT 2>&1 || exit 1
done; done; done
