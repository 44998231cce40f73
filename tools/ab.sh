#!/bin/bash
# Same-box A/B runs (boxes of the pool differ by +-3 %: only alternating runs on ONE box compare builds or options).  Nothing here ever
# writes to the product library dmrg.x_amd/libdmrgx_hip.so: saved builds live under tools/ab/NAME/ (git-ignored, they travel with gpurun),
# engines find their library beside themselves (rpath $ORIGIN), Python harnesses take DMRGX_LIB.
#   tools/ab.sh save NAME                     copy the current build (engine + library) to tools/ab/NAME/
#   tools/ab.sh engines NAME1 NAME2 ...       saved builds on configs[1] (m = 512, 12 sweeps) and configs[3] (m = 2048, warm-up + 1 sweep, gd), twice, alternating
#   tools/ab.sh opts "optsA" "optsB" ...      the current build on configs[3] with several engine option sets (AB_NSWEEPS=1), twice, alternating
#   tools/ab.sh c2opts "optsA" "optsB" ...    the same on configs[1] (12 sweeps: first sweep and the mean of sweeps 3-12)
#   tools/ab.sh env "A=1,B=2" "-" ...         the current build on configs[3] under several environment settings ("-": none)
#   tools/ab.sh bench NAME1 NAME2 ... [-- bench args]   bench.py --no-sweep --no-cpu-baseline with the library of each saved build (DMRGX_LIB), twice, alternating
set -o pipefail
root=$(pwd); cmd=$1; shift
c4="-Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps ${AB_NSWEEPS:-1} -H_eps_type gd"
c2="-Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 12"
report() {      # report DIR LABEL REP NSTEPS
  python3 - "$@" <<'PY'
import json,sys
d,label,rep,n=sys.argv[1],sys.argv[2],sys.argv[3],int(sys.argv[4])
T=json.load(open(d+"/Timings.json")); S=json.load(open(d+"/DMRGSteps.json"))["table"]; run=json.load(open(d+"/DMRGRun.json"))
col={h:i for i,h in enumerate(T["headers"])}
sw={}
for st,tm in zip(S,T["table"]):
    if st[1]!="Sweep": continue
    e=sw.setdefault(int(st[2]),[0,0.0,0]); e[0]+=1; e[1]+=tm[col["Total"]]; e[2]+=int(tm[col["MatMults"]])
ks=sorted(sw); first=sw[ks[0]]; last=sw[ks[-1]]; late=[sw[k] for k in ks[2:]] or [last]
tm=T["table"][-n:]
ph=" ".join("%s %.2f"%(h,1e3*sum(r[col[h]] for r in tm)/len(tm)) for h in ("Total","Enlr","Kron","Diag","Rdms","Rotb") if h in col)
print(f"[{label}] rep{rep}: first sweep {first[0]/first[1]:.2f} sites/s ({first[2]} MatMults)  last {last[0]/last[1]:.2f} ({last[2]})  sweeps 3+ {sum(l[0] for l in late)/sum(l[1] for l in late):.2f}  "
      f"ms/step over the last {len(tm)} steps: {ph}  E={run['GSEnergy']:.10f}  TridFallbacks {run.get('TridFallbacks')}", flush=True)
PY
}
run_engine() {  # run_engine EXE OUTDIR "engine options"
  mkdir -p $2; timeout -k 10 ${AB_TIMEOUT:-600} $1 $3 -data_dir $2/ > $2.log 2>&1 || { tail $2.log; exit 1; }
  rm -f $2/EntanglementSpectra.json $2/Correlations.json $2/KronStats.json
}
case $cmd in
save) mkdir -p tools/ab/$1; cp dmrg.x_amd/dmrgx-square-lattice dmrg.x_amd/libdmrgx_hip.so tools/ab/$1/; echo saved tools/ab/$1;;
engines) out=$root/gpurun_out/ab_engines; for rep in 1 2; do for v in "$@"; do exe=$root/tools/ab/$v/dmrgx-square-lattice
    run_engine $exe $out/c2_$v "$c2"; report $out/c2_$v "$v configs[1]" $rep 280
    [ -n "$AB_SKIP_C4" ] || { run_engine $exe $out/c4_$v "$c4"; report $out/c4_$v "$v configs[3]" $rep 156; }
  done; done;;
opts|c2opts) out=$root/gpurun_out/ab_$cmd; base=$c4; n=156; [ $cmd = c2opts ] && { base=$c2; n=280; }
  for rep in 1 2; do i=0; for o in "$@"; do i=$((i+1)); run_engine $root/dmrg.x_amd/dmrgx-square-lattice $out/v$i "$base $o"; report $out/v$i "$o" $rep $n; done; done;;
env) out=$root/gpurun_out/ab_env; for rep in 1 2; do i=0; for e in "$@"; do i=$((i+1))
    ( [ "$e" = "-" ] || export $(echo $e | tr ',' ' '); run_engine $root/dmrg.x_amd/dmrgx-square-lattice $out/v$i "$c4" ) || exit 1; report $out/v$i "$e" $rep 156; done; done;;
bench) names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" = "--" ] && shift
  for rep in 1 2; do for v in "${names[@]}"; do
    DMRGX_LIB=$root/tools/ab/$v/libdmrgx_hip.so python3 bench.py --no-sweep --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$v] rep$rep: value %.1f MatMults/s  isolated %.1f  frac %.4f  stage1 %.4f ms  stage2 %.4f ms'%(d['value'],d['matmult_isolated_per_s'],r['frac'],r['stage1_ms_per_matmult'],r['stage2_ms_per_matmult']))"
  done; done;;
*) sed -n 2,12p $0;;
esac
