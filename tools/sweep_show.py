import json, sys
cur = None
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep_sched.txt"):
    if l.startswith("=="): cur = l.strip(); continue
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(cur, "val %.1f iso %.1f TF %.2f frac %.3f s1 %.3f s2 %.3f exec/alg %.3f" % (d["value"], d.get("matmult_isolated_per_s", 0), r["achieved"], r["frac"], r["stage1_ms_per_matmult"], r["stage2_ms_per_matmult"], r["flops_exec_per_matmult"] / r["flops_alg_per_matmult"]))
