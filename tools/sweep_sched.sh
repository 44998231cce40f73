#!/bin/bash
# scheduling-parameter sweep for the plan's split-K / tile mix (run on the GPU box)
set -u
mkdir -p gpurun_out
OUT=gpurun_out/sweep_sched.txt
: > $OUT
for W in ${WORKLOADS:-cfg4 cfg2 cfg5}; do
for T in ${TILES:-64}; do
for U in ${UNITS:-8192}; do
for MN in ${MINS:-16}; do
for TP in ${TAPERS:-1}; do
  echo "== $W tiles=$T units=$U min=$MN taper=$TP" >> $OUT
  DMRGX_TILES=$T DMRGX_SPLIT_UNITS=$U DMRGX_SPLIT_MIN=$MN DMRGX_SPLIT_TAPER=$TP timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 32 --warmup 8 >> $OUT 2>&1 || exit 1
done; done; done; done; done
