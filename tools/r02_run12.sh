#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_run12; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_kron.py -x -q -k "davidson" | tail -2
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for keep in 8; do
  mkdir -p $out/k$keep
  DMRGX_GD_KEEP=$keep timeout -k 10 400 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -H_eps_type gd -data_dir $out/k$keep/ > $out/k$keep.log 2>&1 || { tail $out/k$keep.log; exit 1; }
  echo "keep=$keep"; grep "SWEEP DONE\|FINAL" $out/k$keep.log
  rm -f $out/k$keep/EntanglementSpectra.json $out/k$keep/Correlations.json $out/k$keep/KronStats.json
done
