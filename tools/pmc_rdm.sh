#!/bin/bash
# fabric traffic of one Jacobi round (rocprofv3 PMC passes over tools/rdm_bench.py; FETCH_SIZE in KiB of 64-B requests -> x2 on gfx950, WRITE_SIZE x1)
set -u
out=$(pwd)/gpurun_out/pmc_rdm; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for PMC in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $PMC | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $out/$N -o pmc -- python3 $root/tools/rdm_bench.py ${1:-cfg4} > $out/$N.log 2>&1 || echo "pass $N failed"
done
python3 - $out <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for f in sorted(glob.glob(out+'/*/**/*counter_collection.csv',recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'jacobi_round' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(f.split('/')[-3], k, 'per launch mean %.4g (n=%d)'%(sum(v)/len(v),len(v)))
PY
grep "rdm_create" $out/FETCH_SIZE.log | tail -2
