#!/bin/bash
# A/B of tools/ab/lib_old.so vs lib_new.so on one box: kron parity with the new build, then bench lines per workload
set -o pipefail
out=gpurun_out/ab_kron; mkdir -p $out
cp tools/ab/lib_new.so dmrg.x_amd/libdmrgx_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q > $out/kron_tests.log 2>&1; rc=$?
tail -3 $out/kron_tests.log
[ $rc -ne 0 ] && { tail -40 $out/kron_tests.log; exit $rc; }
for rep in 1 2; do for v in old new; do
  cp tools/ab/lib_$v.so dmrg.x_amd/libdmrgx_hip.so
  for W in ${WORKLOADS:-cfg4real cfg3 cfg2}; do
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 48 --warmup 16 > $out/b.json 2>> $out/err.txt || { tail $out/err.txt; exit 1; }
    python3 -c "
import json;d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]);r=d['roofline']
print('$v $W rep$rep value %.1f iso %.1f frac %.4f stage1 %.4f ms stage2 %.4f ms'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult']))"
  done
done; done
cp tools/ab/lib_new.so dmrg.x_amd/libdmrgx_hip.so
