#!/bin/bash
# Host-side HIP API time of an engine run (usage: tools/hip_api_profile.sh engine-options...)
root=$(pwd); out=$root/gpurun_out/hipt; rm -rf $out; mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $out -o t -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > /dev/null 2>&1
cd $root
python3 - <<PY
import csv,glob,json
f=glob.glob("gpurun_out/hipt/**/t_hip_api_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<10: print("%-40s calls %7s total %9.2f ms avg %8.1f us"%(r["Name"][:40],r["Calls"],float(r["TotalDurationNs"])*1e-6,float(r["AverageNs"])*1e-3))
r=json.load(open("gpurun_out/hipt/data/DMRGRun.json")); print("sweep seconds",r["LastSweepSeconds"])
PY
rm -rf $out
