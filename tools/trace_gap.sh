#!/bin/bash
# kernel timeline between the end of one step's RDM eigensolve and the first MatMult of the next step's eigensolve (configs[3], last steps of sweep 1)
set -e
root=$(pwd); out=$root/gpurun_out/trace_gap; mkdir -p $out/data
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- $root/dmrg.x_amd/dmrgx-square-lattice -Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 1 -H_eps_type gd -data_dir $out/data/ > $out/run.log 2>&1
f=$(find $out -name 'trace_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' > $out/gap.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-60000:]
# find the last jacobi_round of a step followed (later) by ggemm<2,2,2,2> runs; take the middle one of the kept window
idx=[i for i,r in enumerate(rows) if 'jacobi_round' in r['Kernel_Name']]
ends=[i for k,i in enumerate(idx) if k+1==len(idx) or idx[k+1]-i>50]
i0=ends[len(ends)//2]
t0=int(rows[i0]['End_Timestamp'])
acc={}
n2=0
for r in rows[i0+1:]:
    name=r['Kernel_Name'].replace('dmrgx::(anonymous namespace)::','').replace('void ','').split('(')[0][:50]
    if 'ritz_precond' in r['Kernel_Name']:
        n2+=1
        if n2>1: break
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print('%9.1f us  +%8.1f  %s'%((int(r['Start_Timestamp'])-t0)/1e3, d, name))
PY
head -150 $out/gap.txt
rm -f $f
