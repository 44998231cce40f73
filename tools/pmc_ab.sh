#!/bin/bash
# L2 / fabric counters of the bench's GEMM launches for tools/ab/lib_old.so vs lib_new.so (rocprofv3 --pmc passes, kernel trace only)
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_ab; mkdir -p $out
for v in old new; do
  cp tools/ab/lib_$v.so dmrg.x_amd/libdmrgx_hip.so
  for PMC in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
    NAME=$(echo $PMC | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $out/${v}_$NAME -o pmc -- python3 bench.py --no-cpu-baseline --no-sweep --steps 12 --warmup 4 > $out/${v}_$NAME.json 2> $out/${v}_$NAME.err || echo "pass $v $NAME failed"
  done
done
cp tools/ab/lib_new.so dmrg.x_amd/libdmrgx_hip.so
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_ab/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            if 'ggemm' in k:
                print(os.path.basename(os.path.dirname(d)), k, {n: '%.4g' % (sum(v) / len(v)) for n, v in c.items()}, 'n', len(next(iter(c.values()))))
PY
