#!/bin/bash
# round-2 GPU session 4: Jacobi sub-solve with rotations computed once per round (A/B inside one box)
set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run4
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q -k "rdm or RDM or density or truncat" > $out/rdm_tests.log 2>&1; rc=$?
tail -3 $out/rdm_tests.log
[ $rc -ne 0 ] && { tail -40 $out/rdm_tests.log; exit $rc; }
exe=$root/dmrg.x_amd/dmrgx-square-lattice
j="-J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5"
for d in 1; do
  mkdir -p $out/c2_$d $out/c4_$d
  DMRGX_JACOBI_DEDUP=$d timeout -k 10 300 $exe -Lx 8 -Ly 4 $j -mwarmup 512 -nsweeps 2 -data_dir $out/c2_$d/ > $out/c2_$d.log 2>&1
  DMRGX_JACOBI_DEDUP=$d timeout -k 10 600 $exe -Lx 20 -Ly 8 $j -mwarmup 2048 -nsweeps 1 -data_dir $out/c4_$d/ > $out/c4_$d.log 2>&1
  python3 - $out $d <<'PY'
import json,sys
o,d=sys.argv[1],sys.argv[2]
for c,n in (("c2",28),("c4",156)):
    tm=json.load(open(f"{o}/{c}_{d}/Timings.json"))["table"]
    run=json.load(open(f"{o}/{c}_{d}/DMRGRun.json"))
    print(f"dedup={d} {c}: last sweep {run['LastSweepSteps']/run['LastSweepSeconds']:.2f} sites/s  E={run['GSEnergy']:.12f}  per step ms: Diag %.2f Rdms %.2f"%(1e3*sum(r[4] for r in tm[-n:])/n,1e3*sum(r[5] for r in tm[-n:])/n))
PY
  rm -f $out/c?_$d/EntanglementSpectra.json $out/c?_$d/Correlations.json $out/c?_$d/KronStats.json
done
