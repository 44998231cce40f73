#!/usr/bin/env python3
"""Print the parts of a bench.py JSON line that the docs quote.  usage: tools/show_bench.py bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "matmult_isolated_per_s")}, "frac %.4f" % r["frac"], "launch ms %.4f" % r["avg_launch_ms"])
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["reference_row_loop"]["value"])
if "sweep" in d:
    s = d["sweep"]
    print({k: v for k, v in s.items() if k not in ("per_sweep", "configs_1", "configs_4", "in_sweep", "config", "e0_config")})
    print(s["per_sweep"])
    print("configs_1", s["configs_1"]["sites_per_s"], s["configs_1"]["matmults_per_s_in_sweep"], s.get("in_sweep"))
    c4 = s.get("configs_4")
    if c4:
        print("configs_4", {k: c4[k] for k in c4 if k not in ("config", "sweep_energies")})
