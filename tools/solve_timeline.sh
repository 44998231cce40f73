#!/bin/bash
# Kernel timeline of a few solver iterations in the middle of the last sweep of an engine run: start (us), idle gap in front, duration, kernel.
# usage: tools/solve_timeline.sh TAG engine-options...
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/tl_$tag; mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' > $out/timeline.txt
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
tr=[i for i,r in enumerate(rows) if "trid_coop" in r[2]]
# the step whose persistent tridiagonalisation is the 40th from the end: walk back from it to the middle of its eigensolve
i1=tr[-40]; i0=tr[-41]
mid=(i0+i1)//2
while "ritz_precond" not in rows[mid][2]: mid+=1
sh=lambda x:x.replace("(anonymous namespace)::","").replace("dmrgx::","").replace("void ","").split("(")[0][-40:]
t0=rows[mid][0]; prev=rows[mid-1][1]
for s,e,n in rows[mid:mid+46]:
    print("%9.1f  gap %6.1f  dur %7.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,sh(n))); prev=max(prev,e)
PY
rm -f $f; cat $out/timeline.txt
