"""Numpy prototype of the round-3 density-matrix eigensolver (csrc/trid.hip, csrc/stedc.hip), written in the exact data
flow of the HIP kernels so that index conventions, the deflation scan, the secular solver and the rotation chains can be
checked on the CPU before they are debugged on a GPU box:

  trid_pipeline   one "launch" per column j: prologue (every workgroup, redundantly): w_{j-1} from y_{j-1}; updated row j;
                  d_j, e_j, v_j, tau_j;  main (own rows): rank-2 update of reflector j-1 fused with y_j = A v_j
  stedc           level-synchronous Cuppen divide and conquer: uniform-depth tree, brute-force ranking, type-1/2 deflation
                  scan, secular roots in shifted coordinates, Loewner weights, U_full with unit columns for deflated
                  poles, rotation chains applied to rows of U_full, Q_new = blockdiag(Q1, Q2) . U_full
  backtransform   X = H_0 .. H_{n-3} Z in blocks of 32 reflectors with S = striu(V^T V) + diag(1/tau)

Developer tool (not imported by the product or the tests).  Run: python tools/proto_trid_dc.py
"""
import numpy as np

EPS = np.finfo(float).eps


def trid_pipeline(A):
    n = A.shape[0]
    As = A.copy()
    VT = np.zeros((n, n)); tau = np.zeros(n); d = np.zeros(n); e = np.zeros(n)
    y_prev = np.zeros(n); v_prev = np.zeros(n); tau_prev = 0.0
    for j in range(n):
        s = y_prev[j:] @ v_prev[j:]
        w = tau_prev * y_prev - (0.5 * tau_prev * tau_prev * s) * v_prev
        r = As[j, j:] - v_prev[j] * w[j:] - w[j] * v_prev[j:]
        d[j] = r[0]
        x = r[1:]
        v = np.zeros(n); tj = 0.0
        if len(x) >= 1:
            alpha = x[0]; sigma = float(np.sum(x[1:] ** 2))
            v[j + 1] = 1.0
            if sigma == 0.0:
                beta = alpha
            else:
                beta = -np.copysign(np.sqrt(alpha * alpha + sigma), alpha)
                tj = (beta - alpha) / beta
                v[j + 2:] = x[1:] / (alpha - beta)
            e[j] = beta
        As[j + 1:, j + 1:] -= np.outer(v_prev[j + 1:], w[j + 1:]) + np.outer(w[j + 1:], v_prev[j + 1:])
        y = np.zeros(n); y[j + 1:] = As[j + 1:, j + 1:] @ v[j + 1:]
        VT[j] = v; tau[j] = tj
        y_prev, v_prev, tau_prev = y, v, tj
    return d, e[:n - 1], VT, tau


def backtransform(VT, tau, Z, nb=32):
    n = Z.shape[0]
    X = Z.copy()
    nref = max(n - 2, 0)
    blocks = [(b, min(b + nb, nref)) for b in range(0, nref, nb)]
    for b0, b1 in reversed(blocks):
        V = VT[b0:b1].T.copy()                       # n x kb, column c = v_{b0+c}
        t = tau[b0:b1].copy()
        for c in range(b1 - b0):
            if t[c] == 0.0: V[:, c] = 0.0; t[c] = 1.0
        S = np.triu(V.T @ V, 1) + np.diag(1.0 / t)
        W = np.linalg.solve(S, V.T @ X)               # back substitution (S upper triangular)
        X -= V @ W
    return X


def secular_root(dl, z2, rho, j):
    """Root j of 1 + rho sum z2_i / (dl_i - lam) in (dl_j, dl_{j+1}) [last: right of dl_{k-1}]: returns (origin index, tau)."""
    k = len(dl)
    if k == 1: return 0, rho * z2[0]
    last = j == k - 1
    if last:
        width = rho * z2.sum(); o = j; p0, p1 = j - 1, j          # psi: poles up to k-2 matched at k-2; phi: the last pole alone (exact)
        dlt = dl - dl[o]; lo, hi = 0.0, width
        g_mid = 1.0 + rho * np.sum(z2 / (dlt - 0.5 * width))
        if g_mid <= 0: lo = 0.5 * width
        else: hi = 0.5 * width
    else:
        width = dl[j + 1] - dl[j]; p0, p1 = j, j + 1
        dm = dl - dl[j]
        g_mid = 1.0 + rho * np.sum(z2 / (dm - 0.5 * width))
        if g_mid >= 0: o = j; dlt = dm; lo, hi = 0.0, 0.5 * width
        else: o = j + 1; dlt = dl - dl[j + 1]; lo, hi = -0.5 * width, 0.0
    tau = 0.5 * (lo + hi)
    p, q = dlt[p0], dlt[p1]
    nl = p0 + 1                       # poles [0, nl) form psi (matched at p), the rest phi (matched at q)
    for it in range(80):
        den = dlt - tau
        terms = rho * z2 / den
        g = 1.0 + terms.sum()
        err = EPS * (8.0 * np.abs(terms).sum() + 1.0)
        if abs(g) <= err: break
        if g > 0: hi = tau            # g is increasing in tau between two poles
        else: lo = tau
        if hi - lo <= 2.0 * EPS * max(abs(lo), abs(hi)): break
        dterms = terms / den
        psi, dpsi = terms[:nl].sum(), dterms[:nl].sum()
        phi, dphi = terms[nl:].sum(), dterms[nl:].sum()
        # "middle way": psi ~ s + a/(p - t), phi ~ r + b/(q - t), value and slope matched at the current point
        a = dpsi * (p - tau) ** 2; sc = psi - dpsi * (p - tau)
        b = dphi * (q - tau) ** 2; rc = phi - dphi * (q - tau)
        c = 1.0 + sc + rc
        # c (p - t)(q - t) + a (q - t) + b (p - t) = 0
        A2 = c; B2 = -(c * (p + q) + a + b); C2 = c * p * q + a * q + b * p
        cand = []
        if A2 == 0.0:
            if B2 != 0.0: cand.append(-C2 / B2)
        else:
            disc = B2 * B2 - 4 * A2 * C2
            if disc >= 0:
                sq = np.sqrt(disc)
                qq = -0.5 * (B2 + np.copysign(sq, B2))
                if qq != 0.0: cand.append(C2 / qq)
                cand.append(qq / A2)
        new = None
        for t in cand:
            if lo < t < hi: new = t; break
        tau = new if new is not None else 0.5 * (lo + hi)
    return o, tau


def merge(D, Qbd, n1, beta):
    """One Cuppen merge.  D: concatenated child eigenvalues, Qbd: block diagonal eigenvector matrix, beta: coupling."""
    n = len(D)
    rho = 2.0 * abs(beta)
    z = np.concatenate([Qbd[n1 - 1, :n1], np.sign(beta) * Qbd[n1, n1:]]) / np.sqrt(2.0) if beta != 0 else np.zeros(n)
    if beta != 0: z = np.where(np.arange(n) < n1, Qbd[n1 - 1, :], np.sign(beta) * Qbd[n1, :]) / np.sqrt(2.0)
    rank = np.array([np.sum((D < D[i]) | ((D == D[i]) & (np.arange(n) < i))) for i in range(n)])
    order = np.empty(n, int); order[rank] = np.arange(n)        # order[s] = original column at sorted position s
    ds = D[order].copy(); zs = z[order].copy()
    tol = 8.0 * EPS * max(np.abs(ds).max(), np.abs(zs).max())
    nondef = []; deflated = []; rots = []                       # rots: (col_pj, col_jj, c, s)
    if rho * np.abs(zs).max() <= tol:
        deflated = list(range(n))
    else:
        pj = -1
        for jj in range(n):
            if rho * abs(zs[jj]) <= tol: deflated.append(jj); continue
            if pj < 0: pj = jj; continue
            s_ = zs[pj]; c_ = zs[jj]; tau = np.hypot(c_, s_); t = ds[jj] - ds[pj]; c_ /= tau; s_ = -s_ / tau
            if abs(t * c_ * s_) <= tol:
                zs[jj] = tau; zs[pj] = 0.0
                rots.append((order[pj], order[jj], c_, s_))
                tt = ds[pj] * c_ * c_ + ds[jj] * s_ * s_
                ds[jj] = ds[pj] * s_ * s_ + ds[jj] * c_ * c_; ds[pj] = tt
                deflated.append(pj); pj = jj
            else:
                nondef.append(pj); pj = jj
        nondef.append(pj)
    k = len(nondef)
    dl = ds[nondef]; zl = zs[nondef]
    lam = np.zeros(k); org = np.zeros(k, int); taus = np.zeros(k)
    for j in range(k):
        org[j], taus[j] = secular_root(dl, zl * zl, rho, j)
        lam[j] = dl[org[j]] + taus[j]
    # dl_i - lam_j in shifted form
    diff = (dl[:, None] - dl[org][None, :]) - taus[None, :]          # [i, j]
    zhat = np.zeros(k)
    for i in range(k):
        w = diff[i, i]
        for j in range(k):
            if j != i: w *= diff[i, j] / (dl[i] - dl[j])
        zhat[i] = np.copysign(np.sqrt(abs(w)), zl[i])
    U = zhat[:, None] / diff if k else np.zeros((0, 0))
    if k: U /= np.linalg.norm(U, axis=0)[None, :]
    vals = np.concatenate([lam, ds[deflated]])
    nn = len(vals)
    frank = np.array([np.sum((vals < vals[i]) | ((vals == vals[i]) & (np.arange(nn) < i))) for i in range(nn)])
    Ufull = np.zeros((n, n))
    for j in range(k):
        Ufull[order[nondef], frank[j]] = U[:, j]
    pcol = {}
    for t, sp in enumerate(deflated):
        Ufull[order[sp], frank[k + t]] = 1.0; pcol[order[sp]] = frank[k + t]
    # rotation chains on the rows of Ufull, in reverse order
    t = len(rots) - 1
    while t >= 0:
        b = t; a = t
        while a > 0 and rots[a - 1][1] == rots[a][0]: a -= 1
        R = Ufull[rots[b][1], :].copy()
        for u in range(b, a - 1, -1):
            cp, cj, c_, s_ = rots[u]
            e = np.zeros(n); e[pcol[cp]] = 1.0
            Ufull[cj, :] = s_ * e + c_ * R
            R = c_ * e - s_ * R
        Ufull[rots[a][0], :] = R
        t = a - 1
    Dn = np.empty(nn); Dn[frank] = vals
    Qn = np.zeros_like(Qbd)
    Qn[:n1, :] = Qbd[:n1, :n1] @ Ufull[:n1, :]
    Qn[n1:, :] = Qbd[n1:, n1:] @ Ufull[n1:, :]
    return Dn, Qn, k


def stedc(d, e, leaf=32):
    n = len(d)
    scale = max(np.abs(d).max(), np.abs(e).max() if n > 1 else 0.0)
    if scale == 0.0: return np.zeros(n), np.eye(n), []
    d = d / scale; e = e / scale
    depth = 0
    while -(-n // (1 << depth)) > leaf: depth += 1
    def bounds(level):              # node boundaries at a depth: repeated halving
        b = [0, n]
        for _ in range(level):
            nb = []
            for lo, hi in zip(b[:-1], b[1:]): nb += [lo, (lo + hi) // 2]
            b = nb + [n]
        return b
    dm = d.copy()
    for lev in range(1, depth + 1):
        for s in bounds(lev)[1:-1]:
            if s in bounds(lev - 1): continue
            dm[s - 1] -= abs(e[s - 1]); dm[s] -= abs(e[s - 1])
    b = bounds(depth)
    Ds = []; Qs = []
    for lo, hi in zip(b[:-1], b[1:]):
        T = np.diag(dm[lo:hi]) + np.diag(e[lo:hi - 1], 1) + np.diag(e[lo:hi - 1], -1)
        w, q = np.linalg.eigh(T); Ds.append(w); Qs.append(q)
    stats = []
    for lev in range(depth - 1, -1, -1):
        bb = bounds(lev)
        nD = []; nQ = []
        for i in range(len(bb) - 1):
            D1, D2, Q1, Q2 = Ds[2 * i], Ds[2 * i + 1], Qs[2 * i], Qs[2 * i + 1]
            n1 = len(D1)
            Qbd = np.zeros((n1 + len(D2),) * 2); Qbd[:n1, :n1] = Q1; Qbd[n1:, n1:] = Q2
            s = bb[i] + n1
            Dn, Qn, k = merge(np.concatenate([D1, D2]), Qbd, n1, e[s - 1])
            nD.append(Dn); nQ.append(Qn); stats.append((len(Dn), k))
        Ds, Qs = nD, nQ
    return Ds[0] * scale, Qs[0], stats


def check(A, name):
    n = A.shape[0]
    d, e, VT, tau = trid_pipeline(A)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    Q = backtransform(VT, tau, np.eye(n))
    print(f"{name}: n={n} trid |Q^T A Q - T|={np.abs(Q.T @ A @ Q - T).max():.2e} |QQ^T-I|={np.abs(Q @ Q.T - np.eye(n)).max():.2e}", end=" ")
    w, Z, stats = stedc(d, e)
    X = backtransform(VT, tau, Z)
    wr = np.linalg.eigvalsh(A)
    print(f"eig err={np.abs(w - wr).max():.2e} (tol {3e-15 * n * np.abs(wr).max():.1e}) orth={np.abs(X.T @ X - np.eye(n)).max():.2e} "
          f"resid={np.abs(X.T @ A @ X - np.diag(w)).max():.2e} top-merge k/n={stats[-1] if stats else None}")


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (3, 5, 33, 70, 130, 300):
        M = rng.standard_normal((n, max(2, n // 2 + 3))); M /= np.linalg.norm(M)
        check(M @ M.T, "gram")
    for n, kk in ((130, 77), (300, 300), (520, 60)):
        U, _ = np.linalg.qr(rng.standard_normal((n, n)))
        s = np.exp(-0.35 * np.arange(n)) * rng.uniform(0.5, 1.0, n); s[kk:] = 0
        A = (U * s ** 2) @ U.T; A = 0.5 * (A + A.T)
        check(A, "graded")
    A = np.eye(40); check(A, "identity")
    A = np.zeros((10, 10)); check(A, "zero")
    T = np.diag(np.ones(64) * 2) - np.diag(np.ones(63), 1) - np.diag(np.ones(63), -1); check(T, "laplace")
    W = np.diag(np.abs(np.arange(-10, 11)).astype(float)) + np.diag(np.ones(20), 1) + np.diag(np.ones(20), -1); check(np.kron(np.eye(2), W), "wilkinson x2")
