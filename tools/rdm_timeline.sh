#!/bin/bash
# Kernel timeline of ONE truncation step of an engine run: every launch between the density-matrix build and the rotation, with the
# idle gap in front of it (usage: tools/rdm_timeline.sh TAG WHICH engine-options...; WHICH = index of the trid_coop launch to show)
set -e
tag=$1; which=$2; shift 2
root=$(pwd)
out=$root/gpurun_out/tl_$tag
mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_trace.csv' | head -1)
python3 - "$f" $which <<'PY' > $out/timeline.txt
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "trid_coop" in r[2] or "trid_step" in r[2]]
firsts=[i for k,i in enumerate(idx) if k==0 or idx[k-1]!=i-1]
i0=firsts[int(sys.argv[2])]
lo=max(0,i0-25); hi=min(len(rows),i0+int(__import__("os").environ.get("TL_SPAN","150")))
t0=rows[lo][0]
prev=rows[lo][0]
for s,e,n in rows[lo:hi]:
    short=n.split("(")[0].replace("dmrgx::","").replace("(anonymous namespace)::","").replace("void ","")[:44]
    print("%9.1f us  gap %7.1f  dur %8.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,short))
    prev=e
PY
rm -f $f
cat $out/timeline.txt
