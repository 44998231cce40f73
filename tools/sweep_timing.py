"""Phase breakdown of an engine run (developer tool): python tools/sweep_timing.py <engine options...>"""
import json, os, subprocess, sys, tempfile
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmrg.x_amd", "dmrgx-square-lattice")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run([exe, *sys.argv[1:], "-data_dir", d + "/"], capture_output=True, text=True)
    print(r.stdout[-600:], r.stderr[-600:])
    t = json.load(open(d + "/Timings.json")); s = json.load(open(d + "/DMRGSteps.json"))
    rows = [dict(zip(t["headers"], x)) for x in t["table"]]
    srows = [dict(zip(s["headers"], x)) for x in s["table"]]
    sw = [r for r, q in zip(rows, srows) if q["LoopType"] == "Sweep"]
    n = len(sw)
    tot = {k: sum(r[k] for r in sw) for k in ("Total", "Enlr", "Kron", "Diag", "Rdms", "Rotb", "MatMults")}
    print(f"sweep steps {n}: total {tot['Total']:.3f} s -> {n / tot['Total']:.1f} sites/s; per step ms: " +
          ", ".join(f"{k} {1e3 * tot[k] / n:.2f}" for k in ("Enlr", "Kron", "Diag", "Rdms", "Rotb")) +
          f"; other {1e3 * (tot['Total'] - sum(tot[k] for k in ('Enlr','Kron','Diag','Rdms','Rotb'))) / n:.2f}; MatMults/step {tot['MatMults'] / n:.1f}")
    big = max(srows, key=lambda q: q["NumStates_H"])
    print("largest superblock:", big["NumStates_H"], "states")
