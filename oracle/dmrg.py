"""Oracle: sweep orchestration, ground-state solve, RDM truncation -- CPU restatement.

Follows include/DMRGBlockContainer.hpp:687-861 (Warmup), :864-993 (Sweeps), :996-1088 (SingleSweep),
:1304-1653 (SingleDMRGStep), :1656-1959 (GetTruncation), :1962-2003 (EigRDM_BlockDiag),
:2006-2057 (FillRotation_BlockDiag).  The SLEPc Krylov-Schur solve (:1488-1499) is third-party code that is
not under /root/reference (slepc-3.8.3): it is replaced here by a dense LAPACK solve (small N) or ARPACK
(scipy eigsh) converged to ~1e-13 -- "parity unpinned" for solver details, see oracle/__init__.py.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .qn import QuantumNumbers, OracleError, OpSm, OpSz, OpSp
from .block import Block
from .kron import KronBlocks, KronEye_Explicit, KronSumConstruct_explicit, KronSumOperator


def lowest_eigenpair(H, seed=0, dense_below=1500, tol=1e-13):
    """EPS_HEP / EPS_SMALLEST_REAL / nev=1 (include/DMRGBlockContainer.hpp:1489-1498)."""
    N = H.shape[0]
    if isinstance(H, KronSumOperator):
        H = H.as_linear_operator()
    elif N <= dense_below:
        w, v = np.linalg.eigh(H.toarray())
        return float(w[0]), v[:, 0].copy()
    rng = np.random.default_rng(seed)
    v0 = rng.standard_normal(N)
    w, v = spla.eigsh(H, k=1, which="SA", tol=tol, v0=v0, ncv=min(N, 24), maxiter=20000)
    return float(w[0]), v[:, 0].copy()


def GetTruncation(kb, psi, MStates):
    """RDM blocks, full spectra, global m-cut, rotation matrices (include/DMRGBlockContainer.hpp:1656-1959).

    Returns dict per side: RotMatT (CSR m x NStates), QN, TruncErr, spectra (list of (sector idx, eigenvalues)).
    """
    if psi.shape[0] != kb.NumStates():
        raise OracleError(1, "Incorrect vector length.")
    Lq, Rq = kb.LeftBlock.Magnetization, kb.RightBlock.Magnetization
    eigen = ([], [])   # entries (eigval, seqIdx, epsIdx, blkIdx)   :82-94
    vecs = ([], [])    # per seqIdx: eigenvector matrix, columns in epsIdx order
    for idx in range(kb.size()):
        Istart, Iend = kb.Offsets(idx), kb.Offsets(idx + 1)
        Idx_L, Idx_R = kb.LeftIdx(idx), kb.RightIdx(idx)
        N_L, N_R = Lq.qn_size[Idx_L], Rq.qn_size[Idx_R]
        if Iend - Istart != N_L * N_R:
            raise OracleError(1, "Incorrect segment length.")
        # PsiT = N_R x N_L column-major view of v[Istart:]  (:1731)  <=>  Psi[l, r] = v[Istart + l*N_R + r]
        Psi = psi[Istart:Iend].reshape(N_L, N_R)
        rdmd = (Psi @ Psi.T, Psi.T @ Psi)  # :1733-1734
        for side, blk_idx in ((0, Idx_L), (1, Idx_R)):
            w, v = np.linalg.eigh(rdmd[side])       # EPSLAPACK, all eigenpairs (:1976-1982)
            w, v = w[::-1], v[:, ::-1]              # EPS_LARGEST_REAL ordering
            for eps_idx in range(w.shape[0]):
                eigen[side].append((float(w[eps_idx]), idx, eps_idx, blk_idx))
            vecs[side].append(v)
    out = []
    for side, q in ((0, Lq), (1, Rq)):
        spectra = [(e[3], e[0]) for e in eigen[side]]
        lst = sorted(eigen[side], key=lambda e: -e[0])       # stable_sort(greater_eigval) :1795
        m = min(MStates, len(lst))                            # :1819-1820
        # test aid (not in the reference): the eigenvalues on both sides of the cut -- a parity comparison of the kept
        # subspace is only well-defined when lam_kept_min is above round-off and separated from lam_dropped_max
        cut = (lst[m - 1][0] if m > 0 else 0.0, lst[m][0] if m < len(lst) else 0.0)
        lst = lst[:m]
        lst = sorted(lst, key=lambda e: e[3])                 # stable_sort(less_blkIdx) :1852
        NStates = q.NumStates()
        rows, cols, vals = [], [], []
        for row, (eigval, seq, eps, blk) in enumerate(lst):   # FillRotation_BlockDiag :2032-2054
            start, n = q.qn_offset[blk], q.qn_size[blk]
            rows.append(np.full(n, row))
            cols.append(np.arange(start, start + n))
            vals.append(vecs[side][seq][:, eps])
        RotMatT = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, NStates))
        trunc = 1.0
        for e in lst:                                         # :1872-1875
            trunc -= (e[0] > 0) * e[0]
        counts = {}
        for e in lst:                                         # :1878-1892
            counts[e[3]] = counts.get(e[3], 0) + 1
        blks = sorted(counts)
        QN = QuantumNumbers([q.qn_list[b] for b in blks], [counts[b] for b in blks])
        out.append(dict(RotMatT=RotMatT, QN=QN, TruncErr=trunc, spectra=spectra, kept=lst, cut=cut))
    return out[0], out[1]


class StepRecord(dict):
    pass


def _block_operator(blk, op, isite):
    if op == OpSz:
        return blk.SzData[isite]
    if op == OpSp:
        return blk.SpData[isite]
    if op == OpSm:
        return blk.SpData[isite].T.tocsr()          # Sm = Sp^T (src/DMRGBlock.cpp:630-632)
    raise OracleError(1, "Correlators take Sz, Sp and Sm operators.")


def operator_product(blk, ops):
    """Product of single-site operators of one block in its basis, identity for an empty list
    (CalculateOperatorProducts, include/DMRGBlockContainer.hpp:2333-2410)."""
    n = blk.NumStates()
    P = sp.identity(n, format="csr")
    for (op, isite) in ops:
        P = P @ _block_operator(blk, op, isite)
    return P.tocsr()


def correlator_value(kb, psi, sys_ops, env_ops):
    """< psi | P_sys (x) P_env | psi > on the KronBlocks (include/DMRGBlockContainer.hpp:2255-2303): with
    Psi[gL, gR] = psi this is sum(Psi * (P_sys Psi P_env^T))."""
    _, _, _, _, _, gL, gR = kb.rows()
    nL, nR = kb.LeftBlock.NumStates(), kb.RightBlock.NumStates()
    Psi = sp.csr_matrix((psi, (gL, gR)), shape=(nL, nR))
    PL, PR = operator_product(kb.LeftBlock, sys_ops), operator_product(kb.RightBlock, env_ops)
    out = (PL @ Psi @ PR.T).tocsr()
    # only components inside the target sector overlap with psi
    return float(Psi.multiply(out).sum())


class DMRGOracle:
    """CPU DMRG following DMRGBlockContainer<Block::SpinBase, J1J2XXZModel_SquareLattice>."""

    def __init__(self, Ham, mwarmup, qn_sector=0.0, seed=1234, verbose=False, matrix_free_above=1 << 62):
        self.Ham = Ham
        self.mwarmup = int(mwarmup)
        self.qn_sector = float(qn_sector)
        self.num_sites = Ham.NumSites()
        self.AddSite = Block.single_site()
        self.sys_blocks = [None] * (self.num_sites - 1)
        self.sys_ninit = 0
        self.steps = []
        self.seed = seed
        self.verbose = verbose
        self.matrix_free_above = matrix_free_above
        self.gse = None
        self.trunc_err = []
        self.GlobIdx = 0
        self.LoopIdx = 0
        self.measurements = []      # (SysOps, EnvOps) with block-local site indices
        self.corr_values = []       # one row per measurement step

    def SetUpCorrelation(self, OpList):
        """include/DMRGBlockContainer.hpp:627-682: operators on the right half are carried to the environment block by
        reflection; a correlator living only on the right half is moved to the system block."""
        N = self.num_sites
        sys_ops, env_ops = [], []
        for (op, idx) in OpList:
            if 0 <= idx < N // 2:
                sys_ops.append((op, idx))
            elif N // 2 <= idx < N:
                env_ops.append((op, N - 1 - idx))
            else:
                raise OracleError(1, "Operator index out of range")
        if not sys_ops:
            sys_ops, env_ops = env_ops, []
        self.measurements.append((sys_ops, env_ops))

    # ---- one step (include/DMRGBlockContainer.hpp:1304-1653) ----
    def SingleDMRGStep(self, SysBlock, EnvBlock, MStates, loop="Sweep", do_measurements=False):
        same = SysBlock is EnvBlock
        SysEnl = KronEye_Explicit(SysBlock, self.AddSite, self.Ham.H(SysBlock.NumSites() + 1))
        EnvEnl = SysEnl if same else KronEye_Explicit(EnvBlock, self.AddSite, self.Ham.H(EnvBlock.NumSites() + 1))
        NumSitesTotal = SysEnl.NumSites() + EnvEnl.NumSites()
        Terms = self.Ham.H(NumSitesTotal)
        kb = KronBlocks(SysEnl, EnvEnl, (self.qn_sector,))
        # (the explicit matrix up to matrix_free_above states, as the reference's -do_shell 0 path; above it the same operator matrix-free)
        H = KronSumConstruct_explicit(kb, Terms) if kb.NumStates() <= self.matrix_free_above else KronSumOperator(kb, Terms)
        gse, psi = lowest_eigenpair(H, seed=self.seed + self.GlobIdx)
        BT_L, BT_R = GetTruncation(kb, psi, MStates)
        if do_measurements and self.measurements:                # :1543
            self.corr_values.append([correlator_value(kb, psi, so, eo) for (so, eo) in self.measurements])
        SysOut = Block.with_sectors(SysEnl.NumSites(), BT_L["QN"].qn_list, BT_L["QN"].qn_size)
        SysOut.RotateOperators(SysEnl, BT_L["RotMatT"])
        if same:
            EnvOut = SysOut
        else:
            EnvOut = Block.with_sectors(EnvEnl.NumSites(), BT_R["QN"].qn_list, BT_R["QN"].qn_size)
            EnvOut.RotateOperators(EnvEnl, BT_R["RotMatT"])
        rec = StepRecord(GlobIdx=self.GlobIdx, LoopIdx=self.LoopIdx, loop=loop,
                         NSites_Sys=SysBlock.NumSites(), NSites_Env=EnvBlock.NumSites(),
                         NSites_SysEnl=SysEnl.NumSites(), NSites_EnvEnl=EnvEnl.NumSites(),
                         NStates_SysEnl=SysEnl.NumStates(), NStates_EnvEnl=EnvEnl.NumStates(),
                         NumStates_H=kb.NumStates(), GSEnergy=gse,
                         TruncErr_Sys=BT_L["TruncErr"], TruncErr_Env=BT_R["TruncErr"],
                         NStates_SysRot=SysOut.NumStates(), NStates_EnvRot=EnvOut.NumStates(),
                         sectors_SysEnl=(SysEnl.Magnetization.qn_list, SysEnl.Magnetization.qn_size),
                         sectors_EnvEnl=(EnvEnl.Magnetization.qn_list, EnvEnl.Magnetization.qn_size),
                         nterms=len(Terms), cut_Sys=BT_L["cut"], cut_Env=BT_R["cut"])
        if self.verbose:
            print(f"  [{loop} {self.GlobIdx}] sys {SysBlock.NumSites()} env {EnvBlock.NumSites()} "
                  f"N_sb {kb.NumStates()} E {gse:.12f} trunc {BT_L['TruncErr']:.3e}")
        self.steps.append(rec)
        self.gse = gse
        self.trunc_err.append(BT_L["TruncErr"])
        self.GlobIdx += 1
        self.last = dict(kb=kb, psi=psi, Terms=Terms, SysEnl=SysEnl, EnvEnl=EnvEnl, BT_L=BT_L, BT_R=BT_R, H=H)
        return SysOut, EnvOut

    # ---- warm-up (include/DMRGBlockContainer.hpp:687-861) ----
    def warmup_schedule(self):
        """(sys_sites, env_sites) of every warm-up step (:809-840)."""
        c = self.Ham.NumEnvSites()
        if c % 2:
            c *= 2
        s, out = c, []
        while s < self.num_sites // 2:
            full = ((s + 2) // c + 1) * c
            env = full - s - 2
            env += ((s - env) // c) * c
            if env < 1 or env > s:
                raise OracleError(1, f"Incorrect number of sites. Got {env}.")
            out.append((s, env))
            s += 1
        return c, out

    def Warmup(self):
        if self.num_sites % 2:
            raise OracleError(1, "Total number of sites must be even.")
        c, sched = self.warmup_schedule()
        self.sys_blocks[0] = Block.single_site()
        self.sys_ninit = 1
        while self.sys_ninit < c:  # :786-790 exact blocks
            prev = self.sys_blocks[self.sys_ninit - 1]
            self.sys_blocks[self.sys_ninit] = KronEye_Explicit(prev, self.AddSite, self.Ham.H(prev.NumSites() + 1))
            self.sys_ninit += 1
        if self.sys_ninit >= self.num_sites // 2:
            raise OracleError(1, "No DMRG Steps were performed since all site operators were created exactly.")
        for (s, env) in sched:
            assert s == self.sys_ninit
            so, eo = self.SingleDMRGStep(self.sys_blocks[s - 1], self.sys_blocks[env - 1], self.mwarmup, loop="Warmup",
                                         do_measurements=(s + 1 == self.num_sites // 2))     # :833
            self.sys_blocks[s] = so
            self.sys_blocks[env] = eo
            self.sys_ninit += 1
        self.LoopIdx += 1

    # ---- sweeps (include/DMRGBlockContainer.hpp:996-1088) ----
    def SingleSweep(self, MStates, min_block=1):
        N = self.num_sites
        self.trunc_err = []
        for iblock in range(N // 2, N - min_block - 2):          # :1040-1055
            insys, inenv, outsys, outenv = iblock - 1, N - iblock - 3, iblock, N - iblock - 2
            so, eo = self.SingleDMRGStep(self.sys_blocks[insys], self.sys_blocks[inenv], MStates)
            self.sys_blocks[outsys], self.sys_blocks[outenv] = so, eo
        for iblock in range(min_block, N // 2):                    # :1059-1074
            insys, inenv, outsys, outenv = N - iblock - 3, iblock - 1, N - iblock - 2, iblock
            so, eo = self.SingleDMRGStep(self.sys_blocks[insys], self.sys_blocks[inenv], MStates,
                                         do_measurements=(outsys == outenv))                  # :1073
            self.sys_blocks[outsys], self.sys_blocks[outenv] = so, eo
        self.LoopIdx += 1

    def Sweeps(self, nsweeps=0, msweeps=()):
        if msweeps:
            for m in msweeps:
                self.SingleSweep(int(m))
        else:
            for _ in range(nsweeps):
                self.SingleSweep(self.mwarmup)
