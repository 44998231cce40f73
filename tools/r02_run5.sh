#!/bin/bash
# round-2 GPU session 5: split-K floor for small superblocks (A/B), kron parity
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run5
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q > $out/kron_tests.log 2>&1; rc=$?
tail -3 $out/kron_tests.log
[ $rc -ne 0 ] && { tail -40 $out/kron_tests.log; exit $rc; }
for w in cfg2 cfg3; do
 for smin in 16 0 8 4; do
  DMRGX_SPLIT_MIN=$smin timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline --steps 96 --warmup 32 --workload $w > $out/b_${w}_$smin.json 2>> $out/bench.err || { tail $out/bench.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('$out/b_${w}_$smin.json').read().strip().splitlines()[-1]);r=d['roofline']
print('$w split_min=$smin value %.1f iso %.1f frac %.4f stage1 %.4f ms stage2 %.4f ms tiles2 %d'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult'],r['tiles_stage2']))"
 done
done
