#!/bin/bash
# A/B of two library builds on the RDM phase: tools/ab/lib_old.so vs lib_new.so (RDM parity with each, then engine sweeps)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/ab_engine; mkdir -p $out
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for v in new old; do
  cp tools/ab/lib_$v.so dmrg.x_amd/libdmrgx_hip.so
  timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q -k "rdm" > $out/rdm_$v.log 2>&1; rc=$?
  tail -2 $out/rdm_$v.log; [ $rc -ne 0 ] && { tail -40 $out/rdm_$v.log; cp tools/ab/lib_old.so dmrg.x_amd/libdmrgx_hip.so; exit $rc; }
  mkdir -p $out/c2_$v $out/c4_$v
  timeout -k 10 300 $exe -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 2 -data_dir $out/c2_$v/ > $out/c2_$v.log 2>&1
  timeout -k 10 600 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 1 -H_eps_type gd -verbose 1 -data_dir $out/c4_$v/ > $out/c4_$v.log 2>&1
  python3 - $out $v <<'PY'
import json,sys,re
o,v=sys.argv[1],sys.argv[2]
for c,n in (("c2",28),("c4",156)):
    tm=json.load(open(f"{o}/{c}_{v}/Timings.json"))["table"]; run=json.load(open(f"{o}/{c}_{v}/DMRGRun.json"))
    print(f"{v} {c}: last sweep {run['LastSweepSteps']/run['LastSweepSeconds']:.2f} sites/s  E={run['GSEnergy']:.10f}  Rdms %.2f ms/step"%(1e3*sum(r[5] for r in tm[-n:])/n))
sw=[int(x) for x in re.findall(r"block-Jacobi sweeps (\d+)", open(f"{o}/c4_{v}.log").read())]
print(f"{v} c4 Jacobi sweeps per step: mean %.2f max %d"%(sum(sw)/len(sw), max(sw)))
PY
  rm -f $out/c?_$v/EntanglementSpectra.json $out/c?_$v/Correlations.json $out/c?_$v/KronStats.json
done
cp tools/ab/lib_old.so dmrg.x_amd/libdmrgx_hip.so
