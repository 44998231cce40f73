#!/bin/bash
# round-2 GPU session 6: scheduler knobs on the cfg4real workload (same box)
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run6
mkdir -p $out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline --steps 32 --warmup 8 > $out/b_$tag.json 2>> $out/bench.err || { tail $out/bench.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('$out/b_$tag.json').read().strip().splitlines()[-1]);r=d['roofline']
print('$tag value %.1f iso %.1f frac %.4f stage1 %.3f ms stage2 %.3f ms tiles %d/%d big %d'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult'],r['tiles_stage1'],r['tiles_stage2'],r['tiles_128']))"
}
run base X=1
run mixed DMRGX_TILES=mixed
run units4k DMRGX_SPLIT_UNITS=4096
run units16k DMRGX_SPLIT_UNITS=16384
run units32k DMRGX_SPLIT_UNITS=32768
run cluster4 DMRGX_CLUSTER=4
run cluster16 DMRGX_CLUSTER=16
run taper DMRGX_SPLIT_TAPER=1
run base2 X=1
