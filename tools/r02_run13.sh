#!/bin/bash
# round-2 GPU session 13: minimum-vertex-cover term groups -- parity, then groups / flops / sweep time against one-sided merging
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_run13; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_engine.py -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log; [ $rc -ne 0 ] && { tail -60 $out/tests.log; exit $rc; }
exe=$root/dmrg.x_amd/dmrgx-square-lattice
for mode in cover onesided; do
  mkdir -p $out/$mode
  if [ $mode = onesided ]; then export DMRGX_MERGE_ONE_SIDED=1; else unset DMRGX_MERGE_ONE_SIDED; fi
  timeout -k 10 400 $exe -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 2 -H_eps_type gd -step_profile 1 -data_dir $out/$mode/ > $out/$mode.log 2>&1 || { tail $out/$mode.log; exit 1; }
  echo "$mode"; grep "SWEEP DONE" $out/$mode.log
  python3 - $out/$mode <<'PY'
import json,sys
ks=[k for k in json.load(open(sys.argv[1]+"/KronStats.json")) if k["LoopType"]=="Sweep"][156:]
print("  sweep 2: mean groups %.1f  mean F_alg %.1f GF  GEMM TF/s %.2f"%(sum(k["n_groups"] for k in ks)/len(ks), sum(k["flops_alg"] for k in ks)/len(ks)/1e9,
      sum(k["flops_alg"]*k["timed_applies"] for k in ks)/(sum(k["ms_stage1"]+k["ms_stage2"] for k in ks)*1e-3)/1e12))
PY
  rm -f $out/$mode/EntanglementSpectra.json $out/$mode/Correlations.json $out/$mode/KronStats.json
done
