#!/bin/bash
# A/B of one engine OPTION on whole runs: tools/ab_opt_sweeps.sh -option "A B ..." engine-options...   (per loop: MatMults, seconds, last energy)
opt=$1; vals=$2; shift 2
root=$(pwd); out=$root/gpurun_out/ab_sweeps; mkdir -p $out
for v in $vals; do
  mkdir -p $out/o$v
  timeout -k 10 900 $root/dmrg.x_amd/dmrgx-square-lattice "$@" $opt $v -data_dir $out/o$v/ > $out/o$v.log 2>&1 || { tail -20 $out/o$v.log; exit 1; }
  python3 - $out/o$v "$opt" $v <<'PY'
import json,sys
d,var,v=sys.argv[1:4]
t=json.load(open(d+"/Timings.json")); s=json.load(open(d+"/DMRGSteps.json"))
h=t["headers"]; mm=[r[h.index("MatMults")] for r in t["table"]]; tot=[r[h.index("Total")] for r in t["table"]]; dg=[r[h.index("Diag")] for r in t["table"]]
sh=s["headers"]; li=[r[sh.index("LoopIdx")] for r in s["table"]]; en=[r[sh.index("GSEnergy")] for r in s["table"]]
for loop in sorted(set(li)):
    idx=[i for i in range(len(mm)) if li[i]==loop]
    print(f"{var} {v} loop {loop}: steps {len(idx)} MatMults {sum(mm[i] for i in idx)} seconds {sum(tot[i] for i in idx):.3f} (solve {sum(dg[i] for i in idx):.3f}) last E {en[idx[-1]]:.12f}")
PY
  rm -f $out/o$v/EntanglementSpectra.json $out/o$v/KronStats.json $out/o$v/Correlations.json
done
