#!/usr/bin/env python3
"""Isolated throughput of the grouped MFMA-f64 GEMM kernel on plain shapes (kernel efficiency without the
superblock's ragged task tables).  Usage (GPU box): python tools/gemm_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd.superblock import dgemm_nn

shapes = [(4096, 4096, 4096), (8192, 8192, 1024), (2048, 2048, 8192), (850, 850, 10000), (512, 512, 512), (1024, 1024, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    A = torch.randn(M, K, dtype=torch.float64, device="cuda")
    B = torch.randn(K, N, dtype=torch.float64, device="cuda")
    C = torch.empty(M, N, dtype=torch.float64, device="cuda")
    dgemm_nn(A, B, C); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        dgemm_nn(A, B, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t0 = time.perf_counter(); R = A @ B; torch.cuda.synchronize(); 
    e0.record(); R = A @ B; e1.record(); torch.cuda.synchronize()
    ms_ref = e0.elapsed_time(e1)
    print(f"{M}x{N}x{K}: dmrgx ggemm {2.0*M*N*K/ms/1e9:7.2f} TF/s ({ms:.3f} ms)   rocBLAS(torch) {2.0*M*N*K/ms_ref/1e9:7.2f} TF/s   maxerr {float((C-R).abs().max()):.2e}")
