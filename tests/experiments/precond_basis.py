"""Experiment (round 4), not a test: python3 tests/experiments/precond_basis.py 6 4 48   (CPU, a few minutes)
Does a block-energy basis make the diagonal preconditioner of the superblock solve better?
For mid-sweep steps of an oracle run: H_sb in (a) the density-matrix eigenbasis of both blocks (what the engine has), (b) the same blocks rotated,
sector by sector, into the eigenbasis of their block Hamiltonian.  Start vector: the exact ground state truncated (density-matrix cut on the
left enlarged block) to a discarded weight ~1e-6 -- the structure of a transformed start vector.  Generalised Davidson, diagonal preconditioner,
||r|| <= 1e-8 |theta|.  Prints MatMults for both."""
import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.hamiltonian import J1J2XXZModel_SquareLattice
from oracle.dmrg import DMRGOracle, GetTruncation, lowest_eigenpair
from oracle.block import Block
from oracle.kron import KronBlocks, KronEye_Explicit, KronSumConstruct_explicit

def energy_basis(blk):
    """Block rotated so that H is diagonal inside every sector."""
    q = blk.Magnetization
    n = q.NumStates()
    H = blk.H.toarray()
    Q = np.zeros((n, n))
    for s in range(len(q.qn_size)):
        a, b = q.qn_offset[s], q.qn_offset[s] + q.qn_size[s]
        w, v = np.linalg.eigh(H[a:b, a:b])
        Q[a:b, a:b] = v
    out = Block.with_sectors(blk.NumSites(), q.qn_list, q.qn_size)
    out.RotateOperators(blk, sp.csr_matrix(Q.T))
    return out

def davidson(H, v0, tol=1e-8, maxit=200):
    d = H.diagonal()
    V = [v0 / np.linalg.norm(v0)]
    W = []
    for it in range(1, maxit + 1):
        W.append(H @ V[-1])
        Vm, Wm = np.array(V).T, np.array(W).T
        G = Vm.T @ Wm
        th, Y = np.linalg.eigh((G + G.T) / 2)
        y = Y[:, 0]
        x = Vm @ y
        r = Wm @ y - th[0] * x
        if np.linalg.norm(r) <= tol * abs(th[0]):
            return it, th[0]
        den = d - th[0]
        den[np.abs(den) < 1e-3] = 1e-3
        t = r / den
        for _ in range(2):
            t -= Vm @ (Vm.T @ t)
        V.append(t / np.linalg.norm(t))
    return maxit, th[0]

def lanczos_like(H, v0, tol=1e-8, maxit=200):
    """same driver without preconditioner (steepest-descent direction = residual): a Lanczos-equivalent count"""
    V = [v0 / np.linalg.norm(v0)]; W = []
    for it in range(1, maxit + 1):
        W.append(H @ V[-1])
        Vm, Wm = np.array(V).T, np.array(W).T
        G = Vm.T @ Wm
        th, Y = np.linalg.eigh((G + G.T) / 2)
        y = Y[:, 0]; x = Vm @ y; r = Wm @ y - th[0] * x
        if np.linalg.norm(r) <= tol * abs(th[0]): return it
        t = r.copy()
        for _ in range(2): t -= Vm @ (Vm.T @ t)
        V.append(t / np.linalg.norm(t))
    return maxit

def start_vector(kb, psi, target=1e-6):
    """psi with the left enlarged block truncated by its density matrix to a discarded weight ~target"""
    n = kb.LeftBlock.Magnetization.NumStates()
    for m in range(n, 0, -1):
        L, R = GetTruncation(kb, psi, m)
        if L["TruncErr"] > target: break
    RT = L["RotMatT"]          # m x n_left
    P = (RT.T @ RT).toarray()
    out = np.zeros_like(psi)
    Lq, Rq = kb.LeftBlock.Magnetization, kb.RightBlock.Magnetization
    for idx in range(kb.size()):
        a, b = kb.Offsets(idx), kb.Offsets(idx + 1)
        il, ir = kb.LeftIdx(idx), kb.RightIdx(idx)
        nl, nr = Lq.qn_size[il], Rq.qn_size[ir]
        lo = Lq.qn_offset[il]
        out[a:b] = (P[lo:lo + nl, lo:lo + nl] @ psi[a:b].reshape(nl, nr)).ravel()
    return out / np.linalg.norm(out), L["TruncErr"], m

if __name__ == "__main__":
    Lx, Ly, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    Hm = J1J2XXZModel_SquareLattice(Lx=Lx, Ly=Ly, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5)
    o = DMRGOracle(Hm, m)
    o.Warmup()
    # wrap the step: at mid-lattice steps of the sweep, run the comparison on the blocks the step was given
    orig = o.SingleDMRGStep
    rows = []
    def wrapped(SysBlock, EnvBlock, MStates, **kw):
        out = orig(SysBlock, EnvBlock, MStates, **kw)
        ns, ne = SysBlock.NumSites(), EnvBlock.NumSites()
        if min(ns, ne) >= Ly and o.last["kb"].NumStates() > 2000:
            kbA, psiA, HA = o.last["kb"], o.last["psi"], o.last["H"]
            v0, te, mk = start_vector(kbA, psiA)
            itA, _ = davidson(HA, v0)
            itL = lanczos_like(HA, v0)
            S2, E2 = energy_basis(SysBlock), energy_basis(EnvBlock)
            SysEnl = KronEye_Explicit(S2, o.AddSite, o.Ham.H(ns + 1))
            EnvEnl = KronEye_Explicit(E2, o.AddSite, o.Ham.H(ne + 1))
            kbB = KronBlocks(SysEnl, EnvEnl, (o.qn_sector,))
            HB = KronSumConstruct_explicit(kbB, o.Ham.H(ns + ne + 2))
            eB, psiB = lowest_eigenpair(HB, seed=1)
            v0B, teB, mkB = start_vector(kbB, psiB)
            itB, _ = davidson(HB, v0B)
            # how diagonal: off-diagonal Frobenius weight
            offA = np.sqrt(max(HA.multiply(HA).sum() - (HA.diagonal() ** 2).sum(), 0)); offB = np.sqrt(max(HB.multiply(HB).sum() - (HB.diagonal() ** 2).sum(), 0))
            rows.append((ns, ne, kbA.NumStates(), itL, itA, itB))
            print(f"sys {ns:3d} env {ne:3d} N {kbA.NumStates():7d}  E {o.gse:.10f} / {eB:.10f}  start err {te:.1e}/{teB:.1e}  MatMults: none {itL:3d}  diag(rho basis) {itA:3d}  diag(energy basis) {itB:3d}   |offdiag| {offA:.1f} -> {offB:.1f}", flush=True)
        return out
    o.SingleDMRGStep = wrapped
    o.SingleSweep(m)
    a = np.array(rows)
    print("mean MatMults: none %.1f  rho-basis diag %.1f  energy-basis diag %.1f" % (a[:, 3].mean(), a[:, 4].mean(), a[:, 5].mean()))
