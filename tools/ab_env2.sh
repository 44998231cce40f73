#!/bin/bash
# like ab_env.sh but each setting may hold several VAR=val separated by commas: tools/ab_env2.sh <workload> "-" "A=1,B=2" ...
W=$1; shift
out=gpurun_out/ab_env; mkdir -p $out
for rep in 1 2; do for E in "$@"; do
  if [ "$E" = "-" ]; then EV=""; else EV="${E//,/ }"; fi
  env $EV timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-sweep --workload $W --steps 48 --warmup 16 > $out/b.json 2>> $out/err.txt || { tail $out/err.txt; }
  python3 -c "
import json;d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]);r=d['roofline']
print('[$E] $W rep$rep value %.1f iso %.1f frac %.4f stage1 %.4f ms stage2 %.4f ms'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult']))"
done; done
