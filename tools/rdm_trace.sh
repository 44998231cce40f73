#!/bin/bash
# Jacobi convergence history of every truncation of an engine run (usage: tools/rdm_trace.sh TAG engine-options...)
tag=$1; shift
out=gpurun_out/rdmtrace_$tag; mkdir -p $out
DMRGX_RDM_TRACE=1 dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/ > $out/run.log 2> $out/trace.log
python3 - $out/trace.log <<'PY'
import sys,re
calls=[];cur=[]
for l in open(sys.argv[1]):
    m=re.search(r"sweep (\d+): max off\^2/total\^2 = ([0-9.e+-]+)",l)
    if not m: continue
    k=int(m.group(1)); v=float(m.group(2))
    if k==0 and cur: calls.append(cur); cur=[]
    cur.append(v)
if cur: calls.append(cur)
for i,c in enumerate(calls): print(i,len(c)-1," ".join("%.0e"%v for v in c))
PY
