#!/bin/bash
# A/B of the start vectors carried through basis overlaps (round 3): MatMults per phase with and without, same energies expected.
# usage: tools/guess_ab.sh engine-options...
root=$(pwd); out=$root/gpurun_out/guess_ab; mkdir -p $out
for v in 0 1; do
  mkdir -p $out/v$v
  timeout -k 10 900 $root/dmrg.x_amd/dmrgx-square-lattice "$@" -wavefunction_guess_overlap $v -data_dir $out/v$v/ > $out/v$v.log 2>&1 || { tail -20 $out/v$v.log; exit 1; }
  python3 - $out/v$v $v <<'PY'
import json,sys
d,v=sys.argv[1:3]
t=json.load(open(d+"/Timings.json")); s=json.load(open(d+"/DMRGSteps.json")); run=json.load(open(d+"/DMRGRun.json"))
h=t["headers"]; mm=[r[h.index("MatMults")] for r in t["table"]]; tot=[r[h.index("Total")] for r in t["table"]]
sh=s["headers"]; lt=[r[sh.index("LoopType")] for r in s["table"]]; li=[r[sh.index("LoopIdx")] for r in s["table"]]; en=[r[sh.index("GSEnergy")] for r in s["table"]]
for loop in sorted(set(li)):
    idx=[i for i in range(len(mm)) if li[i]==loop]
    print(f"overlap={v} loop {loop} ({lt[idx[0]]}): steps {len(idx)} MatMults {sum(mm[i] for i in idx)} seconds {sum(tot[i] for i in idx):.3f} last E {en[idx[-1]]:.12f}")
print("   transformed", run["StartVectorsTransformed"], "through overlap", run["StartVectorsThroughOverlap"], " first-sweep MatMults per step:", [mm[i] for i in range(len(mm)) if li[i]==1][:40])
PY
done
