#!/bin/bash
# A/B of two library builds on the RDM bench (same box): tools/ab/lib_old.so vs lib_new.so
set -u
OUT=gpurun_out/ab_rdm.txt; : > $OUT
cp dmrg.x_amd/libdmrgx_hip.so /tmp/lib_keep.so
for v in old new; do
  cp tools/ab/lib_$v.so dmrg.x_amd/libdmrgx_hip.so
  echo "== $v" >> $OUT
  timeout -k 10 300 python3 tools/rdm_bench.py ${WORKLOADS:-cfg2 cfg3 cfg4} >> $OUT 2>&1 || exit 1
done
cp /tmp/lib_keep.so dmrg.x_amd/libdmrgx_hip.so
cat $OUT
