#!/usr/bin/env python3
"""Time and check dmrgx_rdm_create on the sector tables of a real configs[3] sweep step (workloads.REAL_PROFILES) with a state
whose Schmidt spectrum decays like a DMRG ground state's.  The solver is chosen per process (DMRGX_RDM_SOLVER=jacobi|dc):

    DMRGX_RDM_SOLVER=jacobi python tools/rdm_solver_ab.py ; python tools/rdm_solver_ab.py        # GPU box

Prints the mean create time of one side (what a sweep step asks for) and of both sides, and compares spectra / eigenvectors of
the largest blocks with LAPACK."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from dmrgx_amd import _capi
from dmrgx_amd.superblock import ReducedDensityMatrices
from dmrgx_amd.workloads import REAL_PROFILES, enlarged_sectors, kept_profile

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4real"
decay = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
if name in REAL_PROFILES:
    kl, kr = REAL_PROFILES[name]["left"], REAL_PROFILES[name]["right"]
else:
    kl = kr = kept_profile(int(name))
qn, ls, _ = enlarged_sectors(kl)
rqn, rs, _ = enlarged_sectors(kr)
blocks = [(il, ir) for il in range(len(qn)) for ir in range(len(rqn)) if qn[il] + rqn[ir] == 0.0]
rng = np.random.default_rng(1)
parts = []
for a, b in blocks:
    U, _ = np.linalg.qr(rng.standard_normal((ls[a], ls[a])))
    V, _ = np.linalg.qr(rng.standard_normal((rs[b], rs[b])))
    k = min(ls[a], rs[b])
    s = np.exp(-decay * np.arange(k)) * rng.uniform(0.5, 1.0, k) * np.sqrt(k)
    parts.append((U[:, :k] * s) @ V[:, :k].T)
psi = np.concatenate([p.ravel() for p in parts]); psi /= np.linalg.norm(psi)
d = torch.from_numpy(psi).cuda()
L = _capi.lib()
solver = os.environ.get("DMRGX_RDM_SOLVER", "dc")
print(f"{name}: left sectors {ls} right {rs}; solver {solver}")

def subset(mask_val, reps=5):
    lsz, rsz = (C.c_int32 * len(ls))(*ls), (C.c_int32 * len(rs))(*rs)
    sl, sr = _capi.Sectors(len(ls), lsz), _capi.Sectors(len(rs), rsz)
    bil, bir = (C.c_int32 * len(blocks))(*[b[0] for b in blocks]), (C.c_int32 * len(blocks))(*[b[1] for b in blocks])
    mask = (C.c_uint8 * len(blocks))(*([mask_val] * len(blocks)))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ts = []
    for r in range(reps):
        h = C.c_void_p()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _capi.check(L.dmrgx_rdm_create_subset(C.byref(sl), C.byref(sr), len(blocks), bil, bir, C.c_void_p(d.data_ptr()), C.cast(mask, C.c_void_p), st, C.byref(h)))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        _capi.check(L.dmrgx_rdm_destroy(h))
    return ts

for label, mv in (("left side only", 1), ("both sides", 3)):
    ts = subset(mv)
    print(f"  {label}: create {np.mean(ts[1:])*1e3:.2f} ms (first {ts[0]*1e3:.1f}; runs {[round(t*1e3,2) for t in ts[1:]]})")

rdm = ReducedDensityMatrices(ls, rs, blocks, d)
off = 0
worst = [0.0, 0.0, 0.0]
for k, (a, b) in enumerate(blocks):
    Psi = psi[off:off + ls[a] * rs[b]].reshape(ls[a], rs[b]); off += ls[a] * rs[b]
    for side, rho in ((0, Psi @ Psi.T), (1, Psi.T @ Psi)):
        n = rho.shape[0]
        w_ref = np.linalg.eigvalsh(rho)[::-1]
        w = rdm.eigenvalues(side, k)
        Uv = rdm.eigenvectors(side, k, n).cpu().numpy()
        e = [np.abs(w - w_ref).max() / (n * np.abs(w_ref).max()), np.abs(Uv @ Uv.T - np.eye(n)).max(), np.abs(Uv @ rho @ Uv.T - np.diag(w)).max() / np.linalg.norm(rho)]
        worst = [max(x, y) for x, y in zip(worst, e)]
print(f"  worst over all blocks: |w - w_lapack| / (n |w|max) = {worst[0]:.2e}   |U U^T - I| = {worst[1]:.2e}   |U rho U^T - diag| / |rho|_F = {worst[2]:.2e}")
rdm.destroy()
