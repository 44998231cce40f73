#!/usr/bin/env python3
"""Per-step efficiency of the dominant GEMM inside a sweep, from the KronStats.json of an engine run with -step_profile 1:
tools/kronstats_report.py DIR [last N steps]   (fraction of the 78.6 TF/s f64 peak per step, by position in the lattice)"""
import json, sys
import numpy as np
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 156
ks = [k for k in json.load(open(d + "/KronStats.json")) if k["LoopType"] == "Sweep" and k["timed_applies"] > 0][-n:]
rows = []
for k in ks:
    t = (k["ms_stage1"] + k["ms_stage2"]) / k["timed_applies"] * 1e-3
    rows.append((k["NSites_Sys"], k["n_states"], k["flops_alg"] / 1e9, t * 1e3, k["flops_alg"] / t / 1e12 / 78.6, k["ms_stage1"] / k["timed_applies"], k["ms_stage2"] / k["timed_applies"],
                 k["n_tiles_stage1"], k["n_tiles_stage2"], k["matmults"], k["eigs_seconds"] * 1e3))
for r in rows[::max(1, len(rows) // 40)]:
    print("sys %3d  N %8d  F %6.1f GF  MatMult GEMM %.3f ms  frac %.3f  stage1 %.3f stage2 %.3f  tiles %5d %5d  MatMults %2d  solve %.1f ms" % r)
a = np.array([r[4] for r in rows]); w = np.array([r[2] * r[9] for r in rows])
print("steps %d: mean frac %.3f, flop-weighted %.3f, max %.3f; steps below 0.6: %d holding %.1f %% of the flops" % (len(rows), a.mean(), (a * w).sum() / w.sum(), a.max(), (a < 0.6).sum(), 100 * w[a < 0.6].sum() / w.sum()))
