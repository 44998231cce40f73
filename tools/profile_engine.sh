#!/bin/bash
# rocprofv3 kernel trace of a whole engine run (usage: tools/profile_engine.sh TAG engine-options...)
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/eng_$tag
mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' > $out/summary.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f s"%(tot*1e-9))
for r in rows[:25]:
    print("%-70s calls %7s  total %9.3f ms  avg %9.1f us  %5.1f%%"%(r["Name"][:70],r["Calls"],float(r["TotalDurationNs"])*1e-6,float(r["AverageNs"])*1e-3,float(r["Percentage"])))
PY
rm -f $out/trace_kernel_trace.csv
cat $out/summary.txt
python3 -c "
import json;r=json.load(open('$out/data/DMRGRun.json'));print({k:r[k] for k in r if 'Sweep' in k or 'Energy' in k})"
