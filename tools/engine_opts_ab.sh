#!/bin/bash
# One engine build, several option sets, on configs[3] (warm-up + one sweep): tools/engine_opts_ab.sh "opts A" "opts B" ...   (same box, alternating, twice)
root=$(pwd); out=$root/gpurun_out/opts_ab; mkdir -p $out
base="-Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps ${AB_NSWEEPS:-1} -H_eps_type gd"
for rep in 1 2; do i=0; for o in "$@"; do i=$((i+1)); d=$out/v$i; mkdir -p $d
  timeout -k 10 600 $root/dmrg.x_amd/dmrgx-square-lattice $base $o -data_dir $d/ > $d.log 2>&1 || { tail $d.log; exit 1; }
  python3 - $d "$o" $rep <<'PY'
import json,sys
d,o,rep=sys.argv[1:4]
T=json.load(open(d+"/Timings.json")); tm=T["table"][-156:]; run=json.load(open(d+"/DMRGRun.json"))
ph=" ".join("%s %.2f"%(h,1e3*sum(r[i] for r in tm)/len(tm)) for i,h in enumerate(T["headers"]) if h in ("Total","Diag","Rdms"))
print(f"[{o}] rep{rep}: {run['LastSweepSteps']/run['LastSweepSeconds']:.2f} sites/s  E={run['GSEnergy']:.10f}  ms/step: {ph}  MatMults {run['LastSweepMatMults']}", flush=True)
PY
  rm -f $d/EntanglementSpectra.json $d/Correlations.json $d/KronStats.json
done; done
