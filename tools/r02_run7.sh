#!/bin/bash
# round-2 GPU session 7: BASELINE configs[4] (XY 32x8, m=4096) engine run on ONE GPU + device-memory residency of configs[3]
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run7
mkdir -p $out/cfg5 $out/cfg4
exe=$root/dmrg.x_amd/dmrgx-square-lattice
( while true; do sleep 50; tail -1 $out/cfg5.log 2>/dev/null | cut -c1-100; done ) &
wd=$!
timeout -k 10 1050 $exe -Lx 32 -Ly 8 -J1 1 -Jz1 0 -J2 0 -Jz2 0 -mwarmup 4096 -nsweeps 1 -verbose 1 -data_dir $out/cfg5/ > $out/cfg5.log 2>&1; rc=$?
kill $wd
grep "SWEEP DONE\|FINAL" $out/cfg5.log
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
try:
    run=json.load(open(o+"/cfg5/DMRGRun.json"))
    print({k:run[k] for k in run if k.startswith("Device") or "Sweep" in k or k=="GSEnergy"})
    tm=json.load(open(o+"/cfg5/Timings.json"))["table"]
    n=run["LastSweepSteps"]
    for name,i in (("Total",1),("Diag",4),("Rdms",5),("Rotb",6)):
        print(name, "mean ms/step in sweep: %.1f"%(1e3*sum(r[i] for r in tm[-n:])/n))
except Exception as e: print("cfg5 incomplete:", e)
PY
rm -f $out/cfg5/EntanglementSpectra.json $out/cfg5/Correlations.json
exit $rc
