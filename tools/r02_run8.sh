#!/bin/bash
# round-2 GPU session 8: storage compaction after pruning -- engine parity, residency of configs[3] and configs[4] on one GPU
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run8
mkdir -p $out/cfg5 $out/cfg4
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q > $out/engine_tests.log 2>&1; rc=$?
tail -3 $out/engine_tests.log
[ $rc -ne 0 ] && { tail -40 $out/engine_tests.log; exit $rc; }
exe=$root/dmrg.x_amd/dmrgx-square-lattice
timeout -k 10 600 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -data_dir $out/cfg4/ > $out/cfg4.log 2>&1 || { tail $out/cfg4.log; exit 1; }
grep "SWEEP DONE\|FINAL" $out/cfg4.log
( while true; do sleep 50; tail -1 $out/cfg5.log 2>/dev/null | cut -c1-100; done ) &
wd=$!
timeout -k 10 700 $exe -Lx 32 -Ly 8 -J1 1 -Jz1 0 -J2 0 -Jz2 0 -mwarmup 4096 -nsweeps 1 -data_dir $out/cfg5/ > $out/cfg5.log 2>&1; rc=$?
kill $wd
grep "SWEEP DONE\|FINAL" $out/cfg5.log
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
for c in ("cfg4","cfg5"):
    run=json.load(open(o+"/%s/DMRGRun.json"%c))
    print(c,{k:run[k] for k in run if k.startswith("Device") or "Sweep" in k or k=="GSEnergy"})
    tm=json.load(open(o+"/%s/Timings.json"%c))["table"]
    n=run["LastSweepSteps"]
    print("   per step ms:", {name:round(1e3*sum(r[i] for r in tm[-n:])/n,2) for name,i in (("Total",1),("Enlr",2),("Kron",3),("Diag",4),("Rdms",5),("Rotb",6))})
PY
rm -f $out/cfg?/EntanglementSpectra.json $out/cfg?/Correlations.json $out/cfg?/KronStats.json
exit $rc
