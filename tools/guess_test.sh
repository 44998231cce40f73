for g in 0 1; do
  dmrg.x_amd/dmrgx-square-lattice -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 256 -nsweeps 2 -wavefunction_guess $g -data_dir /tmp/g$g/ > /tmp/g$g.log 2>&1
  echo "guess=$g"; grep "SWEEP DONE\|FINAL" /tmp/g$g.log
done
python3 - <<'PY'
import json
a=json.load(open("/tmp/g0/DMRGSteps.json")); b=json.load(open("/tmp/g1/DMRGSteps.json"))
ta=json.load(open("/tmp/g0/Timings.json")); tb=json.load(open("/tmp/g1/Timings.json"))
ie=a["headers"].index("GSEnergy"); im=ta["headers"].index("MatMults")
worst=max(abs(x[ie]-y[ie])/abs(x[ie]) for x,y in zip(a["table"],b["table"]))
print("max rel energy diff", worst)
print("MatMults/step random:", [r[im] for r in ta["table"]][-30:])
print("MatMults/step guess :", [r[im] for r in tb["table"]][-30:])
PY
