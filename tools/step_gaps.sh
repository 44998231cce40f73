#!/bin/bash
# Where the GPU waits inside ONE sweep step: every launch of a mid-sweep step with the idle gap in front of it, gaps > 8 us marked.
# usage: tools/step_gaps.sh TAG engine-options...
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/gaps_$tag; mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' > $out/gaps.txt
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
tr=[i for i,r in enumerate(rows) if "trid_coop" in r[2]]
i0,i1=tr[-12],tr[-11]
sh=lambda x:x.replace("(anonymous namespace)::","").replace("dmrgx::","").replace("void ","").split("(")[0][-38:]
t0=rows[i0][0]; prev=rows[i0-1][1]
print("one step: %.1f us, %d launches"%((rows[i1][0]-rows[i0][0])/1e3, i1-i0))
for s,e,n in rows[i0:i1]:
    g=(s-prev)/1e3
    print("%9.1f  gap %6.1f%s dur %7.1f  %s"%((s-t0)/1e3,g," <<<" if g>8 else "    ",(e-s)/1e3,sh(n))); prev=max(prev,e)
PY
rm -f $f; cat $out/gaps.txt
