#!/bin/bash
# round-2 GPU session 1: parity tier, pruning A/B (energies must not change), per-step plan statistics of the cfg4 sweep
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/r02_run1
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
exe=$root/dmrg.x_amd/dmrgx-square-lattice
j1j2="-J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5"
for p in 1 0; do
  mkdir -p $out/ab$p
  timeout -k 10 300 $exe -Lx 6 -Ly 4 $j1j2 -mwarmup 96 -nsweeps 2 -prune_ops $p -data_dir $out/ab$p/ > $out/ab$p.log 2>&1
  grep "SWEEP DONE\|FINAL" $out/ab$p.log
done
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
a=json.load(open(o+"/ab1/DMRGSteps.json"))["table"]; b=json.load(open(o+"/ab0/DMRGSteps.json"))["table"]
assert len(a)==len(b)
bad=[(x[0],x[-1],y[-1]) for x,y in zip(a,b) if x!=y]
print("prune A/B: %d steps, %d rows differ"%(len(a),len(bad)), bad[:3])
ta=json.load(open(o+"/ab1/Timings.json"))["table"]; tb=json.load(open(o+"/ab0/Timings.json"))["table"]
print("rotation seconds: pruned %.4f, unpruned %.4f"%(sum(r[6] for r in ta), sum(r[6] for r in tb)))
PY
mkdir -p $out/cfg4
timeout -k 10 900 $exe -Lx 20 -Ly 8 $j1j2 -mwarmup 2048 -nsweeps 2 -step_profile 1 -verbose 1 -data_dir $out/cfg4/ > $out/cfg4.log 2>&1
grep "SWEEP DONE\|FINAL" $out/cfg4.log
rm -f $out/cfg4/EntanglementSpectra.json $out/cfg4/Correlations.json
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
ks=json.load(open(o+"/cfg4/KronStats.json"))
tm=json.load(open(o+"/cfg4/Timings.json"))["table"]
sw=[k for k in ks if k["LoopType"]=="Sweep"]
f=sum(k["flops_alg"]*k["timed_applies"] for k in sw); t=sum(k["ms_stage1"]+k["ms_stage2"] for k in sw)*1e-3
print("sweep steps %d: mean n_states %.3e  mean flops_alg %.2f GF  GEMM TF/s in sweep %.2f (frac %.3f)"%(len(sw),sum(k["n_states"] for k in sw)/len(sw),sum(k["flops_alg"] for k in sw)/len(sw)/1e9,f/t/1e12,f/t/1e12/78.6))
mid=sw[len(sw)//2]
print("mid-sweep step:",{k:mid[k] for k in mid if k not in("sys_qn","env_qn")})
for name,i in (("Enlr",2),("Kron",3),("Diag",4),("Rdms",5),("Rotb",6)):
    print(name, "mean ms/step in sweeps: %.2f"%(1e3*sum(r[i] for r in tm[-len(sw):])/len(sw)))
PY
