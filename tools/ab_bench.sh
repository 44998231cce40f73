#!/bin/bash
# A/B of two library builds on the same GPU box (device-to-device spread is several percent, so versions are only
# comparable inside one call): tools/ab/lib_old.so vs tools/ab/lib_new.so are copied over the in-tree library in turn.
set -u
OUT=gpurun_out/ab.txt; : > $OUT
for rep in 1 2; do for v in old new; do
  cp tools/ab/lib_$v.so dmrg.x_amd/libdmrgx_hip.so
  for W in ${WORKLOADS:-cfg4}; do
    echo "== $v $W rep$rep" >> $OUT
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 48 --warmup 8 >> $OUT 2>&1 || exit 1
  done
done; done
cp tools/ab/lib_new.so dmrg.x_amd/libdmrgx_hip.so
