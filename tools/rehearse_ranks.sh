#!/bin/bash
# Functional rehearsal of the N-rank engine at full size on ONE GPU (host-staged communicator: ranks share the GPU, so the time means nothing):
# tools/rehearse_ranks.sh N engine-options...   -> energies / MatMults of the N-rank run next to the one-rank run
n=$1; shift
root=$(pwd); out=$root/gpurun_out/rehearse; rm -rf $out; mkdir -p $out/r1 $out/rn
timeout -k 10 900 $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/r1/ > $out/r1.log 2>&1 || { tail $out/r1.log; exit 1; }
for r in $(seq 0 $((n-1))); do
  RANK=$r WORLD_SIZE=$n LOCAL_RANK=0 DMRGX_COMM=shm DMRGX_SHM_NAME=dmrgx_rehearse_$$ DMRGX_SHM_MB=256 timeout -k 10 1100 $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/rn/ > $out/rn_$r.log 2>&1 &
  pids="$pids $!"
done
rc=0; for p in $pids; do wait $p || rc=1; done
[ $rc = 0 ] || { tail -5 $out/rn_*.log; exit 1; }
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
a=json.load(open(o+"/r1/DMRGRun.json")); b=json.load(open(o+"/rn/DMRGRun.json"))
sa=json.load(open(o+"/r1/DMRGSteps.json")); sb=json.load(open(o+"/rn/DMRGSteps.json"))
h=sa["headers"]; ie=h.index("GSEnergy"); it=h.index("TruncErr_Sys")
worst=max(abs(x[ie]-y[ie])/abs(x[ie]) for x,y in zip(sa["table"],sb["table"]))
print("1 rank : E %.12f MatMults %d sweep %.2f s ranks %d"%(a["GSEnergy"],a["MatMults"],a["LastSweepSeconds"],a["Ranks"]))
print("N ranks: E %.12f MatMults %d sweep %.2f s ranks %d  TridLaunchPathCalls %s TridFallbacks %s"%(b["GSEnergy"],b["MatMults"],b["LastSweepSeconds"],b["Ranks"],b.get("TridLaunchPathCalls"),b.get("TridFallbacks")))
print("steps %d / %d; largest relative energy difference over all steps %.2e; max TruncErr %.2e"%(len(sa["table"]),len(sb["table"]),worst,max(x[it] for x in sa["table"])))
PY
