#!/bin/bash
# round-2 GPU session 9: folded partial-dot reductions for small superblocks (A/B by env) + parity
set -o pipefail
out=gpurun_out/r02_run9; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_engine.py -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { tail -40 $out/tests.log; exit $rc; }
for rep in 1 2; do for f in 0 1; do
  for W in cfg2 cfg3; do
    DMRGX_EIGS_FOLD=$f timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 96 --warmup 32 > $out/b.json 2>> $out/err.txt || { tail $out/err.txt; exit 1; }
    python3 -c "
import json;d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]);r=d['roofline']
print('fold=$f $W rep$rep value %.1f iso %.1f frac %.4f'%(d['value'],d['matmult_isolated_per_s'],r['frac']))"
  done
done; done
