#!/bin/bash
# per-step breakdown of the eigensolve (GEMM stages by HIP events vs. solver wall) for one engine run: tools/step_breakdown.sh TAG engine-options...
set -o pipefail
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/steps_$tag; mkdir -p $out
timeout -k 10 900 $root/dmrg.x_amd/dmrgx-square-lattice "$@" -step_profile 1 -data_dir $out/ > $out/run.log 2>&1 || { tail $out/run.log; exit 1; }
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
st=json.load(open(o+'/KronStats.json'))['steps']
sw=[s for s in st if s['LoopType']=='Sweep'][-156:]
tot=lambda k:sum(s[k] for s in sw)
mm=tot('matmults'); gem=tot('ms_stage1')+tot('ms_stage2'); eig=1e3*tot('eigs_seconds'); diag=1e3*sum(s['timings_s']['Diag'] for s in sw)
print(f"last sweep: {len(sw)} steps, {mm} MatMults ({mm/len(sw):.1f}/step); per step: Diag {diag/len(sw):.1f} ms = solver {eig/len(sw):.1f} (GEMM stages {gem/len(sw):.1f}, other {(eig-gem)/len(sw):.1f}) + outside solver {(diag-eig)/len(sw):.1f}")
print(f"per MatMult: GEMM {gem/mm:.3f} ms, solver-other {(eig-gem)/mm:.3f} ms")
for k in ('Total','Enlr','Kron','Diag','Rdms','Rotb'): print(k, '%.2f'%(1e3*sum(s['timings_s'][k] for s in sw)/len(sw)), end='  ')
print()
PY
rm -f $out/EntanglementSpectra.json $out/Correlations.json
