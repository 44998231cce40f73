#!/bin/bash
# Host-side wall-clock profile of an engine run (GPU box): tools/hostprof.sh TAG engine-options...
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/hostprof_$tag; mkdir -p $out/data
g++ -O2 -shared -fPIC tools/hostprof.cpp -o tools/libhostprof.so -lrt
timeout -k 10 900 env HOSTPROF_OUT=$out/samples.txt LD_PRELOAD=$root/tools/libhostprof.so $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
python3 tools/hostprof_report.py $out/samples.txt 60 > $out/report.txt
python3 tools/hostprof_report.py $out/samples.txt 60 dmrgx_eigs_lowest,GetTruncation,CalculateCorrelations > $out/report_glue.txt
python3 tools/hostprof_report.py $out/samples.txt 40 rdm_create_impl GetTruncation > $out/report_truncation.txt
rm -f $out/samples.txt $out/data/EntanglementSpectra.json $out/data/Correlations.json $out/data/KronStats.json
cat $out/report.txt
