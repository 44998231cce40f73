"""Experiment (round 5), not a test: python3 tests/experiments/gd_correction.py 6 4 48   (CPU, a few minutes)
Which correction vector should the generalized-Davidson iteration of csrc/eigs.hip expand its basis with?  One MatMult is 1/15 of a
configs[3] solve, so a correction that saves one is worth more than any kernel tuning left.  For the mid-sweep superblocks of an oracle
run (start vector = ground state with a truncation-shaped error, ||r|| <= 1e-8 |theta|, basis 8, GD+1 restart): MatMults with
  diag    t = r / (diag H - theta)                                        (what the library does)
  olsen   t = M^-1 r - eps M^-1 u,  eps = (u . M^-1 r) / (u . M^-1 u)      (Olsen's correction: t orthogonal to u in the M^-1 metric)
  none    t = r                                                           (unpreconditioned: block Lanczos-like)
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from precond_basis import start_vector
from oracle.hamiltonian import J1J2XXZModel_SquareLattice
from oracle.dmrg import DMRGOracle


def gd(H, d, v0, kind, mmax=8, keep=1, tol=1e-8, maxit=300, floor=1e-3):
    V = (v0 / np.linalg.norm(v0))[:, None]
    W = np.zeros((len(v0), 0))
    yprev = None
    for it in range(1, maxit + 1):
        W = np.hstack([W, (H @ V[:, -1])[:, None]])
        G = V.T @ W
        th, Y = np.linalg.eigh((G + G.T) / 2)
        y = Y[:, 0]
        x = V @ y
        r = W @ y - th[0] * x
        if np.linalg.norm(r) <= tol * abs(th[0]):
            return it
        den = d - th[0]
        den[np.abs(den) < floor] = floor
        if kind == "diag":
            t = r / den
        elif kind == "olsen":
            a, b = r / den, x / den
            t = a - (x @ a) / (x @ b) * b
        else:
            t = r.copy()
        if V.shape[1] >= mmax:
            Q = Y[:, :keep]
            if yprev is not None:
                p = np.concatenate([yprev, [0.0]])
                p -= Q @ (Q.T @ p)
                if np.linalg.norm(p) > 1e-8:
                    Q = np.hstack([Q, (p / np.linalg.norm(p))[:, None]])
            V, W = V @ Q, W @ Q
            yprev = None
        else:
            yprev = y
        for _ in range(2):
            t -= V @ (V.T @ t)
        V = np.hstack([V, (t / np.linalg.norm(t))[:, None]])
    return maxit


if __name__ == "__main__":
    Lx, Ly, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    Hm = J1J2XXZModel_SquareLattice(Lx=Lx, Ly=Ly, J1=1.0, Jz1=1.0, J2=0.5, Jz2=0.5)
    o = DMRGOracle(Hm, m)
    o.Warmup()
    kinds = ["diag", "olsen", "none"]
    rows = []
    orig = o.SingleDMRGStep

    def wrapped(SysBlock, EnvBlock, MStates, **kw):
        out = orig(SysBlock, EnvBlock, MStates, **kw)
        if min(SysBlock.NumSites(), EnvBlock.NumSites()) >= Ly and o.last["kb"].NumStates() > 2000:
            kb, psi, H = o.last["kb"], o.last["psi"], o.last["H"]
            v0, te, _ = start_vector(kb, psi)
            d = H.diagonal()
            rows.append([gd(H, d.copy(), v0, k) for k in kinds])
            print(kb.NumStates(), f"diag spread {d.max() - d.min():.2f}", rows[-1], flush=True)
        return out
    o.SingleDMRGStep = wrapped
    o.SingleSweep(m)
    a = np.array(rows, dtype=float)
    for k, mean in zip(kinds, a.mean(axis=0)):
        print(f"{k:6s}: mean MatMults {mean:5.2f}")
