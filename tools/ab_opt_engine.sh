#!/bin/bash
# A/B of engine option sets on configs[3] (warm-up + 2 sweeps, gd): tools/ab_opt_engine.sh "optsA" "optsB" ...
set -o pipefail
root=$(pwd); out=$root/gpurun_out/ab_opt; mkdir -p $out
exe=$root/dmrg.x_amd/dmrgx-square-lattice
i=0
for opts in "$@"; do
  i=$((i+1)); mkdir -p $out/r$i
  timeout -k 10 900 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -H_eps_type gd $opts -data_dir $out/r$i/ > $out/r$i.log 2>&1 || { tail $out/r$i.log; exit 1; }
  python3 - $out/r$i "$opts" <<'PY'
import json,sys
o,opts=sys.argv[1:3]
tm=json.load(open(o+"/Timings.json")); run=json.load(open(o+"/DMRGRun.json")); hdr=tm['headers']; rows=tm['table'][-156:]
f=lambda k:1e3*sum(r[hdr.index(k)] for r in rows)/len(rows)
print(f"[{opts}] sweep 2: {run['LastSweepSteps']/run['LastSweepSeconds']:.2f} sites/s  MatMults {run['LastSweepMatMults']}  E={run['GSEnergy']:.10f}  per step: Total {f('Total'):.1f} Diag {f('Diag'):.1f} Rdms {f('Rdms'):.1f}")
PY
  rm -f $out/r$i/EntanglementSpectra.json $out/r$i/Correlations.json $out/r$i/KronStats.json
done
