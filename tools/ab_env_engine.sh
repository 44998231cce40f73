#!/bin/bash
# A/B of one environment switch on the engine: tools/ab_env_engine.sh VAR A B  (RDM parity once, then 20x8 m=2048 warm-up + 1 sweep with VAR=A and VAR=B)
set -o pipefail
var=$1; shift
root=$(pwd); out=$root/gpurun_out/ab_env; mkdir -p $out
exe=$root/dmrg.x_amd/dmrgx-square-lattice
timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q -k "rdm and not few_sweeps" > $out/rdm.log 2>&1 || { tail -40 $out/rdm.log; exit 1; }
tail -1 $out/rdm.log
for v in "$@"; do
  mkdir -p $out/c4_$v
  env $var=$v timeout -k 10 600 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 1 -H_eps_type gd -verbose 1 -data_dir $out/c4_$v/ > $out/c4_$v.log 2>&1 || { tail $out/c4_$v.log; exit 1; }
  python3 - $out $v $var <<'PY'
import json,sys,re
o,v,var=sys.argv[1:4]
tm=json.load(open(f"{o}/c4_{v}/Timings.json"))["table"]; run=json.load(open(f"{o}/c4_{v}/DMRGRun.json")); n=156
sw=[int(x) for x in re.findall(r"block-Jacobi sweeps (\d+)", open(f"{o}/c4_{v}.log").read())]
print(f"{var}={v}: last sweep {run['LastSweepSteps']/run['LastSweepSeconds']:.2f} sites/s  E={run['GSEnergy']:.10f}  Rdms %.2f ms/step  Jacobi sweeps mean %.2f max %d"%(1e3*sum(r[5] for r in tm[-n:])/n, sum(sw)/len(sw), max(sw)))
PY
  rm -f $out/c4_$v/EntanglementSpectra.json $out/c4_$v/Correlations.json $out/c4_$v/KronStats.json
done
