#!/bin/bash
# round-2 GPU session 11: generalized Davidson option -- parity, then MatMults per sweep and sweep time against Lanczos on configs[3]
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_run11; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kron.py -x -q -k "diag or davidson or eigs" > $out/t1.log 2>&1; rc=$?
tail -3 $out/t1.log; [ $rc -ne 0 ] && { tail -60 $out/t1.log; exit $rc; }
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -x -q -k "davidson" > $out/t2.log 2>&1; rc=$?
tail -3 $out/t2.log; [ $rc -ne 0 ] && { tail -60 $out/t2.log; exit $rc; }
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for t in krylovschur gd; do
  mkdir -p $out/$t
  timeout -k 10 400 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -H_eps_type $t -data_dir $out/$t/ > $out/$t.log 2>&1 || { tail $out/$t.log; exit 1; }
  echo "$t"; grep "SWEEP DONE" $out/$t.log
  rm -f $out/$t/EntanglementSpectra.json $out/$t/Correlations.json $out/$t/KronStats.json
done
