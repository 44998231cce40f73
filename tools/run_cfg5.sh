#!/bin/bash
# configs[4] of BASELINE.json on one GPU: XY 32x8, m=4096, warm-up + 1 sweep (the NNN terms drop out with Jz2 = 0, as in the reference)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/cfg5; mkdir -p $out
timeout -k 10 1000 $root/dmrg.x_amd/dmrgx-square-lattice -Lx 32 -Ly 8 -J1 1 -Jz1 0 -J2 1 -Jz2 0 -mwarmup 4096 -nsweeps 1 -H_eps_type gd -data_dir $out/ > $out/run.log 2>&1 || { tail $out/run.log; exit 1; }
python3 - $out <<'PY'
import json,sys
o=sys.argv[1]
tm=json.load(open(o+"/Timings.json")); run=json.load(open(o+"/DMRGRun.json")); hdr=tm['headers']; n=run['LastSweepSteps']; rows=tm['table'][-n:]
f=lambda k:1e3*sum(r[hdr.index(k)] for r in rows)/len(rows)
print(f"cfg5 sweep: {n} steps in {run['LastSweepSeconds']:.1f} s = {n/run['LastSweepSeconds']:.2f} sites/s  MatMults {run['LastSweepMatMults']}  E={run['GSEnergy']:.8f}")
print(f"per step: Total {f('Total'):.1f} Diag {f('Diag'):.1f} Rdms {f('Rdms'):.1f} Rotb {f('Rotb'):.1f}  resident {run['DeviceBytesResidentAfterSweep']/1e9:.1f} GB peak {run['DeviceBytesPeak']/1e9:.1f} GB")
PY
rm -f $out/EntanglementSpectra.json $out/Correlations.json
