#!/bin/bash
# round-2 GPU session 10: engine RCCL start-up test; Krylov subspace size vs MatMults per sweep on configs[3]
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_run10; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -k "rccl or communicator" > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -ne 0 ] && { tail -40 $out/tests.log; exit $rc; }
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for ncv in 16 24 20; do
  mkdir -p $out/ncv$ncv
  timeout -k 10 400 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -H_eps_ncv $ncv -data_dir $out/ncv$ncv/ > $out/ncv$ncv.log 2>&1 || { tail $out/ncv$ncv.log; exit 1; }
  echo "ncv=$ncv"; grep "SWEEP DONE" $out/ncv$ncv.log
  rm -f $out/ncv$ncv/EntanglementSpectra.json $out/ncv$ncv/Correlations.json $out/ncv$ncv/KronStats.json
done
