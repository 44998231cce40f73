#!/bin/bash
# one GPU session: the whole parity tier (incl. the multi-rank rehearsals), then the default bench line
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/check
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; rc=$?
tail -25 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','config')})
print(d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_row_loop']['value'])
print({k:v for k,v in d['sweep'].items() if k not in ('per_sweep','configs_1','in_sweep')})
print(d['sweep']['configs_1'])"
