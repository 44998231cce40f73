#!/bin/bash
# engine sweeps of the CURRENT build under several environment settings (comma-separated VAR=val; "-" = none): tools/ab_env_engines.sh "-" "A=1" ...
set -o pipefail
root=$(pwd); out=$root/gpurun_out/ab_env_engines; mkdir -p $out
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for rep in 1 2; do for E in "$@"; do
  if [ "$E" = "-" ]; then EV=""; else EV="${E//,/ }"; fi
  tag=$(echo "$E" | tr -c 'A-Za-z0-9=' '_')
  mkdir -p $out/c2_$tag $out/c4_$tag
  env $EV timeout -k 10 300 $exe -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 12 -data_dir $out/c2_$tag/ > $out/c2_$tag.log 2>&1 || { tail $out/c2_$tag.log; exit 1; }
  [ -n "$AB_SKIP_C4" ] || env $EV timeout -k 10 600 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 1 -H_eps_type gd -data_dir $out/c4_$tag/ > $out/c4_$tag.log 2>&1 || { tail $out/c4_$tag.log; exit 1; }
  python3 - $out $tag $rep "$E" <<'PY'
import json,sys,os
o,v,rep,E=sys.argv[1:5]
for c,n in (("c2",280),("c4",156)):
    if not os.path.exists(f"{o}/{c}_{v}/Timings.json"): continue
    T=json.load(open(f"{o}/{c}_{v}/Timings.json")); tm=T["table"]; run=json.load(open(f"{o}/{c}_{v}/DMRGRun.json"))
    ph=" ".join("%s %.2f"%(h,1e3*sum(r[i] for r in tm[-n:])/n) for i,h in enumerate(T["headers"]) if h in ("Total","Diag","Rdms"))
    print(f"[{E}] rep{rep} {c}: {n / sum(r[1] for r in tm[-n:]):.2f} sites/s  E={run['GSEnergy']:.10f}  ms/step: {ph}  MatMults {run['LastSweepMatMults']}")
PY
  rm -f $out/c?_$tag/EntanglementSpectra.json $out/c?_$tag/Correlations.json $out/c?_$tag/KronStats.json
done; done
