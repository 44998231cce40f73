#!/bin/bash
# A/B of whole builds (engine + library) on engine sweeps: tools/ab_engines.sh NAME1 NAME2 ...  with tools/ab/NAME/{dmrgx-square-lattice,libdmrgx_hip.so}
# (save a build with: tools/ab_engines.sh save NAME).  Runs configs[1] (m = 512, 12 sweeps) and configs[3] (m = 2048, 1 sweep), twice each, alternating.
set -o pipefail
root=$(pwd)
if [ "$1" = save ]; then mkdir -p tools/ab/$2; cp dmrg.x_amd/dmrgx-square-lattice dmrg.x_amd/libdmrgx_hip.so tools/ab/$2/; echo saved tools/ab/$2; exit 0; fi
out=$root/gpurun_out/ab_engines; mkdir -p $out
for rep in 1 2; do for v in "$@"; do
  exe=$root/tools/ab/$v/dmrgx-square-lattice
  mkdir -p $out/c2_$v $out/c4_$v
  timeout -k 10 300 $exe -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 12 -data_dir $out/c2_$v/ > $out/c2_$v.log 2>&1 || { tail $out/c2_$v.log; exit 1; }
  [ -n "$AB_SKIP_C4" ] || timeout -k 10 600 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 1 -H_eps_type gd -data_dir $out/c4_$v/ > $out/c4_$v.log 2>&1 || { tail $out/c4_$v.log; exit 1; }
  python3 - $out $v $rep <<'PY'
import json,sys,os
o,v,rep=sys.argv[1:4]
for c,n in (("c2",280),("c4",156)):      # configs[1]: mean over the last 10 of 12 sweeps (one sweep is 0.09 s: too short to compare builds)
    if not os.path.exists(f"{o}/{c}_{v}/Timings.json"): continue
    T=json.load(open(f"{o}/{c}_{v}/Timings.json")); tm=T["table"]; run=json.load(open(f"{o}/{c}_{v}/DMRGRun.json"))
    ph=" ".join("%s %.2f"%(h,1e3*sum(r[i] for r in tm[-n:])/n) for i,h in enumerate(T["headers"]) if h in ("Total","Enlr","Kron","Diag","Rdms","Rotb"))
    rate = n / sum(r[1] for r in tm[-n:])
    print(f"[{v}] rep{rep} {c}: {rate:.2f} sites/s over the last {n} steps (last sweep {run['LastSweepSteps']/run['LastSweepSeconds']:.2f})  E={run['GSEnergy']:.10f}  ms/step: {ph}  MatMults {run['LastSweepMatMults']}")
PY
  rm -f $out/c?_$v/EntanglementSpectra.json $out/c?_$v/Correlations.json $out/c?_$v/KronStats.json
done; done
