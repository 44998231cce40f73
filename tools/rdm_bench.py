#!/usr/bin/env python3
"""Time dmrgx_rdm_create (RDM build + batched block-Jacobi + Rayleigh quotients) on the sector layout of a BASELINE
config with a random normalised psi.  Usage (GPU box): python tools/rdm_bench.py cfg2 cfg4"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd.superblock import ReducedDensityMatrices
from dmrgx_amd.workloads import synthetic_superblock, CONFIGS, kept_profile, enlarged_sectors

for name in sys.argv[1:] or ["cfg2"]:
    kept = kept_profile(CONFIGS[name]["m"])
    qn, sizes, sub = enlarged_sectors(kept)
    blocks = [(il, ir) for il in range(len(qn)) for ir in range(len(qn)) if qn[il] + qn[ir] == 0.0]
    n = sum(sizes[a] * sizes[b] for a, b in blocks)
    psi = torch.randn(n, dtype=torch.float64, device="cuda")
    psi /= psi.norm()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rdm = ReducedDensityMatrices(sizes, sizes, blocks, psi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tr = sum(rdm.eigenvalues(0, k).sum() for k in range(len(blocks)))
        print(f"{name}: N_sb={n} max sector {max(sizes)}  rdm_create {dt*1e3:.1f} ms  sweeps {rdm.sweeps}  trace-1 = {tr-1:.1e}")
        rdm.destroy()
