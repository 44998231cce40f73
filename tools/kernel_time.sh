#!/bin/bash
# developer aid (usage: [KERNELS=trid_coop,dc_leaf] [SKIP_C4=1] tools/kernel_time.sh NAME...; builds saved with tools/ab.sh save): total time of the named kernels in a configs[1] / configs[3] engine run of each saved build (tools/ab/NAME)
set -o pipefail
root=$(pwd); cd /tmp && export TMPDIR=/tmp
c2="-Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 6"
c4="-Lx 20 -Ly 8 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 2048 -nsweeps 0 -H_eps_type gd"
for rep in 1 2; do for v in "$@"; do for cfg in c2 c4; do
  [ $cfg = c2 ] && o="$c2" || o="$c4"
  [ $cfg = c4 ] && [ -n "$SKIP_C4" ] && continue
  out=$root/gpurun_out/tt_$v; rm -rf $out; mkdir -p $out/data
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- $root/tools/ab/$v/dmrgx-square-lattice $o -data_dir $out/data/ > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
  f=$(find $out -name 't_kernel_stats.csv' | head -1)
  python3 - "$f" "$v" $cfg $rep <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    import os
    short=r["Name"].replace("(anonymous namespace)::","").replace("dmrgx::","").split("(")[0]
    if any(k in short for k in os.environ.get("KERNELS","trid_coop").split(",")):
        print(f"[{sys.argv[2]} {sys.argv[3]}] rep{sys.argv[4]}: {short} calls {r['Calls']}  total {float(r['TotalDurationNs'])/1e6:.3f} ms  avg {float(r['AverageNs'])/1e3:.1f} us", flush=True)
PY
  rm -rf $out
done; done; done
