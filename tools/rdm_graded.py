#!/usr/bin/env python3
"""Convergence of the block-Jacobi eigensolver on a density matrix with an exponentially decaying (DMRG-like) spectrum.
Usage (GPU box): DMRGX_RDM_TRACE=1 python tools/rdm_graded.py [decay] [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd.superblock import ReducedDensityMatrices
decay = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(0)
U, _ = np.linalg.qr(rng.standard_normal((n, n)))
V, _ = np.linalg.qr(rng.standard_normal((n, n)))
s = np.exp(-decay * np.arange(n))
s = np.maximum(s, 1e-18)
Psi = (U * s) @ V.T
psi = Psi.ravel() / np.linalg.norm(Psi)
rdm = ReducedDensityMatrices([n], [n], [(0, 0)], torch.from_numpy(psi).cuda())
w = rdm.eigenvalues(0, 0)
ref = np.sort((s / np.linalg.norm(s)) ** 2)[::-1]
print(f"decay {decay} n {n}: sweeps {rdm.sweeps}, max abs eigenvalue error {np.abs(w - ref).max():.2e}")
