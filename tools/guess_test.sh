for g in "0 0" "1 0" "1 1"; do
  set -- $g
  dmrg.x_amd/dmrgx-square-lattice -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup ${M:-256} -nsweeps ${NS:-3} -wavefunction_guess $1 -rdm_warm_start $2 -data_dir /tmp/g$1$2/ > /tmp/g$1$2.log 2>&1
  echo "guess=$1 rdm_warm=$2"; grep "SWEEP DONE\|FINAL" /tmp/g$1$2.log
done
python3 - <<'PY'
import json
a=json.load(open("/tmp/g00/DMRGSteps.json")); b=json.load(open("/tmp/g11/DMRGSteps.json"))
ie=a["headers"].index("GSEnergy"); it=a["headers"].index("TruncErr_Sys")
print("max rel energy diff", max(abs(x[ie]-y[ie])/abs(x[ie]) for x,y in zip(a["table"],b["table"])), "max abs trunc diff", max(abs(x[it]-y[it]) for x,y in zip(a["table"],b["table"])))
for tag in ("00","11"):
    t=json.load(open(f"/tmp/g{tag}/Timings.json")); h=t["headers"]
    rows=t["table"][-28:]
    print(tag, "last sweep per-step ms:", {k: round(1e3*sum(r[h.index(k)] for r in rows)/len(rows),2) for k in ("Diag","Rdms","Rotb","Total")})
PY
