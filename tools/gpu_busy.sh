#!/bin/bash
# GPU-busy fraction of the last sweep of an engine run, and where the idle time sits (usage: tools/gpu_busy.sh TAG NSTEPS engine-options...)
set -e
tag=$1; nsteps=$2; shift 2
root=$(pwd); out=$root/gpurun_out/busy_$tag; mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice "$@" -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_trace.csv' | head -1)
python3 - "$f" $nsteps <<'PY' > $out/busy.txt
import csv,sys,collections
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); n=int(sys.argv[2])
tr=[i for i,r in enumerate(rows) if "trid_coop" in r[2]]
firsts=[i for k,i in enumerate(tr) if k==0 or rows[tr[k-1]][0] < rows[i][0]-1e6]      # one marker per step (rounds of one call are < 1 ms apart)
i0=firsts[-n]; lo=rows[i0][0]; hi=rows[-1][1]
busy=0; gaps=collections.Counter(); prev_end=lo; prev_name="(start)"
for s,e,nm in rows[i0:]:
    busy+=e-s
    g=s-prev_end
    if g>20000:
        short=lambda x:x.replace("(anonymous namespace)::","").replace("dmrgx::","").replace("void ","").split("(")[0][-40:]
        gaps[(short(prev_name),short(nm))]+=g
    prev_end=max(prev_end,e); prev_name=nm
print("last %d steps: wall %.3f s, GPU busy %.3f s = %.1f %%, idle per step %.2f ms"%(n,(hi-lo)/1e9,busy/1e9,100*busy/(hi-lo),(hi-lo-busy)/1e6/n))
print("idle gaps > 20 us by (kernel before -> kernel after), ms per step:")
for k,v in gaps.most_common(14): print("  %6.3f  %s -> %s"%(v/1e6/n,k[0],k[1]))
kt=collections.Counter(); kc=collections.Counter(); small=0
for s,e,nm in rows[i0:]:
    k=nm.replace("(anonymous namespace)::","").replace("dmrgx::","").replace("void ","").split("(")[0][-44:]
    kt[k]+=e-s; kc[k]+=1
    g=s-prev_end
print("kernel time of the last %d steps, ms per step (launches per step):"%n)
for k,v in kt.most_common(28): print("  %7.3f  (%6.1f)  %s"%(v/1e6/n,kc[k]/n,k))
gsum=0; prev_end=lo
for s,e,nm in rows[i0:]:
    g=s-prev_end
    if 0<g<=20000: gsum+=g
    prev_end=max(prev_end,e)
print("gaps <= 20 us between consecutive kernels: %.3f ms per step over %.0f launches per step"%(gsum/1e6/n,len(rows[i0:])/n))
PY
rm -f $f $out/data/EntanglementSpectra.json $out/data/KronStats.json $out/data/Correlations.json
cat $out/busy.txt
