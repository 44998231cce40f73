#!/usr/bin/env python3
"""Projection of the MatMult's strong scaling from ONE GPU: build the striped plan of every rank of a world of W (same superblock),
apply each on this GPU in turn, and report per rank the algorithmic flops, tiles and the apply time.  t(world 1) / max_r t(rank r) is
what the GEMM stages alone would give on W GPUs; the all-gather of x (printed as bytes) and the Lanczos vector work come on top.
It measures load balance and tile granularity of the stripe rule -- not RCCL.   usage (GPU box): tools/stripe_projection.py [workload] [W ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd.superblock import KronPlan
from dmrgx_amd.workloads import synthetic_superblock

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4real"
worlds = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
sb = synthetic_superblock(name)
out = {"workload": name, "n_states": sb.n_states, "worlds": {}}
t1 = None
for W in worlds:
    rows = []
    for r in range(W):
        plan = KronPlan(sb, world_size=W, rank=r)
        I = plan.info
        x = torch.randn(I.vec_len, dtype=torch.float64, device="cuda")
        y = torch.zeros(I.vec_len, dtype=torch.float64, device="cuda")
        yl = y[I.local_offset:I.local_offset + I.local_len]
        for _ in range(3):
            plan.apply(x, yl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.apply(x, yl)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        rows.append({"rank": r, "flops_alg_GF": I.flops_alg / 1e9, "tiles": I.n_tiles_stage1 + I.n_tiles_stage2, "local_len": I.local_len, "apply_ms": ms})
        plan.destroy()
    tmax = max(x["apply_ms"] for x in rows)
    if W == 1:
        t1 = tmax
    out["worlds"][W] = {"ranks": rows, "max_apply_ms": tmax, "sum_flops_GF": sum(x["flops_alg_GF"] for x in rows),
                        "projected_gemm_speedup": (t1 / tmax) if t1 else None, "allgather_bytes_per_matmult": 8 * sb.n_states}
    print(f"W={W}: max apply {tmax:.3f} ms, min {min(x['apply_ms'] for x in rows):.3f} ms, sum flops {out['worlds'][W]['sum_flops_GF']:.1f} GF"
          + (f", projected GEMM speed-up {t1 / tmax:.2f}x" if t1 else ""), flush=True)
json.dump(out, open(os.path.join("gpurun_out", f"stripe_projection_{name}.json"), "w"), indent=1)
