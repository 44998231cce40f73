#!/bin/bash
# round-2 GPU session 3: borrowed spectra for unread blocks (engine parity + cfg4 sweep), two-deep operand prefetch A/B
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run3
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q > $out/engine_tests.log 2>&1; rc=$?
tail -5 $out/engine_tests.log
[ $rc -ne 0 ] && { tail -40 $out/engine_tests.log; exit $rc; }
DMRGX_GG_DEEP=1 timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q > $out/kron_tests_deep.log 2>&1; rc=$?
tail -3 $out/kron_tests_deep.log
[ $rc -ne 0 ] && { tail -40 $out/kron_tests_deep.log; exit $rc; }
for d in 0 1 0 1; do
  DMRGX_GG_DEEP=$d timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline --steps 48 --warmup 16 > $out/bench_deep$d.json 2>> $out/bench.err || { tail $out/bench.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('$out/bench_deep$d.json').read().strip().splitlines()[-1]);r=d['roofline']
print('deep=$d value %.1f iso %.1f frac %.4f stage1 %.3f ms stage2 %.3f ms'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult']))"
done
exe=$root/dmrg.x_amd/dmrgx-square-lattice
mkdir -p $out/cfg4
timeout -k 10 900 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -data_dir $out/cfg4/ > $out/cfg4.log 2>&1
grep "SWEEP DONE\|FINAL" $out/cfg4.log
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
tm=json.load(open(o+"/cfg4/Timings.json"))["table"]
n=156
for name,i in (("Total",1),("Enlr",2),("Kron",3),("Diag",4),("Rdms",5),("Rotb",6)):
    print(name, "mean ms/step in last sweep: %.2f"%(1e3*sum(r[i] for r in tm[-n:])/n))
PY
rm -f $out/cfg4/EntanglementSpectra.json $out/cfg4/Correlations.json $out/cfg4/KronStats.json
