#!/bin/bash
# bench lines for several library builds on one box: tools/ab_libs.sh <workload> lib1.so lib2.so ...   (two rounds, interleaved)
W=$1; shift
out=gpurun_out/ab_libs; mkdir -p $out
cp dmrg.x_amd/libdmrgx_hip.so $out/keep.so
for rep in 1 2; do for L in "$@"; do
  cp tools/ab/$L dmrg.x_amd/libdmrgx_hip.so
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 48 --warmup 16 > $out/b.json 2>> $out/err.txt || { tail $out/err.txt; }
  python3 -c "
import json;d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]);r=d['roofline']
print('$L $W rep$rep value %.1f iso %.1f frac %.4f stage1 %.4f ms stage2 %.4f ms'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult']))"
done; done
cp $out/keep.so dmrg.x_amd/libdmrgx_hip.so
