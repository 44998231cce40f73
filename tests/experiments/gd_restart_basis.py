"""Experiment (round 5), not a test: python3 tests/experiments/gd_restart_basis.py 6 4 48   (CPU, a few minutes)
How small can the search space of the generalized-Davidson solve be?  The vector work of an iteration is 5 j + 7 passes over a basis of j
vectors (csrc/eigs.hip), so a small maximal basis makes it a small constant -- if the restart does not cost MatMults.  For the mid-sweep
superblocks of an oracle run (start vector = ground state with a truncation-shaped error of ~1e-6, ||r|| <= 1e-8 |theta|): MatMults
for a maximal basis of mmax vectors, restarted to the `keep` lowest Ritz vectors plus (GD+1) the previous iteration's Ritz vector."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from precond_basis import start_vector
from oracle.hamiltonian import J1J2XXZModel_SquareLattice
from oracle.dmrg import DMRGOracle


def gd(H, d, v0, mmax, keep, plus, tol=1e-8, maxit=300):
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
        den[np.abs(den) < 1e-3] = 1e-3
        t = r / den
        if V.shape[1] >= mmax:
            Q = Y[:, :keep]
            if plus and yprev is not None:
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
    variants = [(24, 12, False), (16, 8, False), (12, 4, True), (10, 3, True), (8, 4, False), (8, 3, True), (8, 2, True), (8, 1, True), (6, 2, True), (6, 1, True), (5, 1, True), (4, 1, True), (3, 1, True)]
    rows = []
    orig = o.SingleDMRGStep

    def wrapped(SysBlock, EnvBlock, MStates, **kw):
        out = orig(SysBlock, EnvBlock, MStates, **kw)
        if min(SysBlock.NumSites(), EnvBlock.NumSites()) >= Ly and o.last["kb"].NumStates() > 2000:
            kb, psi, H = o.last["kb"], o.last["psi"], o.last["H"]
            v0, te, _ = start_vector(kb, psi)
            d = H.diagonal()
            rows.append([gd(H, d.copy(), v0, *v) for v in variants])
            print(kb.NumStates(), rows[-1], flush=True)
        return out
    o.SingleDMRGStep = wrapped
    o.SingleSweep(m)
    a = np.array(rows, dtype=float)
    for v, mean in zip(variants, a.mean(axis=0)):
        # passes over one vector per iteration: (4 j + 6) with the dots fused into the correction kernel, j averaged over the cycle, + restart
        lo = v[1] + (1 if v[2] else 0) + 1
        javg = (lo + v[0]) / 2
        print(f"mmax {v[0]:2d} keep {v[1]} {'+1' if v[2] else '  '}: mean MatMults {mean:5.2f}   ~{4 * javg + 6:.0f} vector passes per iteration")
