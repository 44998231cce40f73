#!/bin/bash
# configs[1] (J1-J2 8x4, m = 512) with several option sets, alternating, twice: tools/c2_opts_ab.sh "opts A" "opts B" ...; first sweep and the mean of sweeps 3-12
root=$(pwd); out=$root/gpurun_out/c2_ab; mkdir -p $out
base="-Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 12"
for rep in 1 2; do i=0; for o in "$@"; do i=$((i+1)); d=$out/v$i; mkdir -p $d
  timeout -k 10 300 $root/dmrg.x_amd/dmrgx-square-lattice $base $o -data_dir $d/ > $d.log 2>&1 || { tail $d.log; exit 1; }
  python3 - $d "$o" $rep <<'PY'
import json,sys
d,o,rep=sys.argv[1:4]
T=json.load(open(d+"/Timings.json")); S=json.load(open(d+"/DMRGSteps.json"))["table"]; run=json.load(open(d+"/DMRGRun.json"))
col={h:i for i,h in enumerate(T["headers"])}
sw={}
for st,tm in zip(S,T["table"]):
    if st[1]!="Sweep": continue
    e=sw.setdefault(int(st[2]),[0,0.0,0]); e[0]+=1; e[1]+=tm[col["Total"]]; e[2]+=int(tm[col["MatMults"]])
ks=sorted(sw); first=sw[ks[0]]; late=[sw[k] for k in ks[2:]]
tm=T["table"][-280:]
ph=" ".join("%s %.2f"%(h,1e3*sum(r[col[h]] for r in tm)/len(tm)) for h in ("Total","Diag","Rdms"))
print(f"[{o}] rep{rep}: first sweep {first[0]/first[1]:.1f} sites/s ({first[2]} MatMults)  sweeps 3-12 {sum(l[0] for l in late)/sum(l[1] for l in late):.1f} sites/s ({late[-1][2]} MatMults)  ms/step {ph}  E={run['GSEnergy']:.10f}", flush=True)
PY
  rm -f $d/EntanglementSpectra.json $d/Correlations.json $d/KronStats.json
done; done
