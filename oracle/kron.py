"""Oracle: sector-pair (KronBlock) algebra of the superblock.

Restates include/DMRGKron.hpp:117-213 (KronBlocks_t), :501-656 (KronBlocksIterator),
src/DMRGKron.cpp:52-456 (MatKronEyeConstruct), :459-615 (KronEye_Explicit), :759-841 (KronSumConstruct),
:891-989 (KronSumGetSubmatrices), :1340-1477 (KronSumFillMatrix), :1706-1824 (KronSumSetUpShellTerms) and
:1827-1869 (MatMult_KronSumShell).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import scipy.sparse as sp

from .qn import OpSm, OpSz, OpSp, OpEye, OracleError
from .block import Block, csr_from_rows, _sorted

KS_TOL = 1.0e-16  # include/DMRGKron.hpp:396


class KronBlocks:
    """Ordered list of (QN, IL, IR, size) tuples (include/DMRGKron.hpp:124-213)."""

    def __init__(self, LeftBlock, RightBlock, QNSectors=()):
        self.LeftBlock, self.RightBlock = LeftBlock, RightBlock
        Lq, Rq = LeftBlock.Magnetization, RightBlock.Magnetization
        kb = []
        qset = set(float(q) for q in QNSectors)
        for IL in range(Lq.NumSectors()):
            for IR in range(Rq.NumSectors()):
                qn = Lq.qn_list[IL] + Rq.qn_list[IR]
                if len(qset) == 0 or qn in qset:  # exact float compare, as :165
                    kb.append((qn, IL, IR, Lq.qn_size[IL] * Rq.qn_size[IR]))
        if len(qset) == 0:
            kb.sort(key=lambda t: -t[0])  # python's sort is stable == std::stable_sort(DescendingQN) :157
        self.kb = kb
        self.kb_size = [t[3] for t in kb]
        self.kb_offset = [0]
        for s in self.kb_size:
            self.kb_offset.append(self.kb_offset[-1] + s)
        self.kb_map = {(t[1], t[2]): i for i, t in enumerate(kb)}
        self.num_states = self.kb_offset[-1]

    def size(self):
        return len(self.kb)

    def NumStates(self):
        return self.num_states

    def Map(self, il, ir):
        return self.kb_map.get((il, ir), -1)

    def Offsets(self, *a):
        if len(a) == 1:
            return self.kb_offset[a[0]]
        i = self.Map(*a)
        return self.kb_offset[i] if i >= 0 else -1

    def LeftIdx(self, k):
        return self.kb[k][1]

    def RightIdx(self, k):
        return self.kb[k][2]

    def rows(self):
        """Decode every superblock row like KronBlocksIterator (include/DMRGKron.hpp:587-620).

        Returns arrays (k, IL, IR, locL, locR, globL, globR), one entry per row.
        """
        Lq, Rq = self.LeftBlock.Magnetization, self.RightBlock.Magnetization
        ks, ILs, IRs, lL, lR, gL, gR = [], [], [], [], [], [], []
        for k, (_, IL, IR, size) in enumerate(self.kb):
            nR = Rq.qn_size[IR]
            loc = np.arange(size)
            ks.append(np.full(size, k))
            ILs.append(np.full(size, IL))
            IRs.append(np.full(size, IR))
            lL.append(loc // nR)
            lR.append(loc % nR)
            gL.append(Lq.qn_offset[IL] + loc // nR)
            gR.append(Rq.qn_offset[IR] + loc % nR)
        cat = lambda v: np.concatenate(v) if v else np.zeros(0, dtype=np.int64)
        return tuple(cat(v) for v in (ks, ILs, IRs, lL, lR, gL, gR))


# --------------------------------------------------------------------------------------------------
#  Block enlargement
# --------------------------------------------------------------------------------------------------
def _merged_sectors(kb):
    """Equal-QN KronBlocks are merged into one output sector (src/DMRGKron.cpp:561-574)."""
    qn_list, qn_size, last = [], [], 0.0
    for (qn, _, _, size) in kb.kb:
        if qn < last or len(qn_list) == 0:
            qn_list.append(qn)
            qn_size.append(size)
        else:
            qn_size[-1] += size
        last = qn
    return qn_list, qn_size


def kron_eye_rows_literal(LeftBlock, RightBlock, kb, side, op_type, isite):
    """Literal row loop of MatKronEyeConstruct for one operator (src/DMRGKron.cpp:323-437).

    side 0: O_L (x) 1 ; side 1: 1 (x) O_R.  Returns {row: [(col, val), ...]} of the output operator.
    """
    Lq, Rq = LeftBlock.Magnetization, RightBlock.Magnetization
    blk = (LeftBlock, RightBlock)[side]
    mat = (blk.SzData if op_type == OpSz else blk.SpData)[isite]
    rows = {}
    for k, (_, IL, IR, size) in enumerate(kb.kb):
        nR = Rq.qn_size[IR]
        if op_type == OpSz:  # :377-382
            col_NStatesR, fws_O = nR, kb.kb_offset[k]
        else:  # :383-389  (+1 on the block index of the side the operator lives on)
            if side == 0:
                col_NStatesR, fws_O = Rq.Sizes(IR), kb.Offsets(IL + 1, IR)
            else:
                col_NStatesR, fws_O = Rq.Sizes(IR + 1), kb.Offsets(IL, IR + 1)
        if fws_O == -1:
            continue
        if side == 0:
            bks, flg = Lq.OpBlockToGlobalRangeStart(IL, op_type)
        else:
            bks, flg = Rq.OpBlockToGlobalRangeStart(IR, op_type)
        if not flg:
            continue
        for loc in range(size):
            locL, locR = loc // nR, loc % nR
            Irow = kb.kb_offset[k] + loc
            if side == 1:  # :409-416
                r = Rq.qn_offset[IR] + locR
                a, b = mat.indptr[r], mat.indptr[r + 1]
                ent = [(locL * col_NStatesR + (int(c) - bks) + fws_O, float(v))
                       for c, v in zip(mat.indices[a:b], mat.data[a:b])]
            else:  # :417-424
                r = Lq.qn_offset[IL] + locL
                a, b = mat.indptr[r], mat.indptr[r + 1]
                ent = [((int(c) - bks) * col_NStatesR + locR + fws_O, float(v))
                       for c, v in zip(mat.indices[a:b], mat.data[a:b])]
            if ent:
                rows[Irow] = ent
    return rows


def _kron_eye_blockwise(LeftBlock, RightBlock, kb, side, op_type, isite, nout):
    """Same operator as kron_eye_rows_literal, assembled sector-block by sector-block (vectorised)."""
    Lq, Rq = LeftBlock.Magnetization, RightBlock.Magnetization
    blk = (LeftBlock, RightBlock)[side]
    mat = (blk.SzData if op_type == OpSz else blk.SpData)[isite]
    R, C, V = [], [], []
    for k, (_, IL, IR, size) in enumerate(kb.kb):
        ILc, IRc = (IL + op_type, IR) if side == 0 else (IL, IR + op_type)
        kc = kb.Map(ILc, IRc)
        if kc < 0:
            continue
        if side == 0:
            sub = mat[Lq.qn_offset[IL]:Lq.qn_offset[IL + 1], Lq.qn_offset[ILc]:Lq.qn_offset[ILc + 1]]
            big = sp.kron(sub, sp.identity(Rq.qn_size[IR], format="csr"), format="coo")
        else:
            sub = mat[Rq.qn_offset[IR]:Rq.qn_offset[IR + 1], Rq.qn_offset[IRc]:Rq.qn_offset[IRc + 1]]
            big = sp.kron(sp.identity(Lq.qn_size[IL], format="csr"), sub, format="coo")
        R.append(big.row + kb.kb_offset[k])
        C.append(big.col + kb.kb_offset[kc])
        V.append(big.data)
    if not R:
        return csr_from_rows(nout, {})
    m = sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(nout, nout))
    return _sorted(m)


def KronEye_Explicit(LeftBlock, RightBlock, Terms, literal=False):
    """Combine two blocks: O_i(x)1, 1(x)o_j, merged sectors, H_out (src/DMRGKron.cpp:459-615)."""
    LeftBlock.CheckOperatorBlocks()   # :511-516
    RightBlock.CheckOperatorBlocks()
    kb = KronBlocks(LeftBlock, RightBlock, ())
    nL, nR = LeftBlock.NumSites(), RightBlock.NumSites()
    nout = nL + nR
    qn_list, qn_size = _merged_sectors(kb)
    out = Block.with_sectors(nout, qn_list, qn_size)
    if out.NumStates() != kb.NumStates():
        raise OracleError(1, "Mismatch in number of states.")
    N = out.NumStates()
    for side, blk, shift in ((0, LeftBlock, 0), (1, RightBlock, nL)):
        for isite in range(blk.NumSites()):
            for op_type, dst in ((OpSz, out.SzData), (OpSp, out.SpData)):
                if literal:
                    dst[isite + shift] = csr_from_rows(N, kron_eye_rows_literal(LeftBlock, RightBlock, kb, side, op_type, isite))
                else:
                    dst[isite + shift] = _kron_eye_blockwise(LeftBlock, RightBlock, kb, side, op_type, isite, N)
    for t in Terms:  # :607-610
        if t.Isite >= nout or t.Jsite >= nout:
            raise OracleError(1, "Term indices must be less than nsites_out")
    out.H = KronSumConstruct_explicit(kb, Terms)  # :612
    return out


# --------------------------------------------------------------------------------------------------
#  KronSum: term classification shared by the explicit and the shell path
# --------------------------------------------------------------------------------------------------
def classify_terms(kb, Terms):
    """Keep inter-block terms, drop a==0, reflect right sites (src/DMRGKron.cpp:788-807)."""
    nL, nR = kb.LeftBlock.NumSites(), kb.RightBlock.NumSites()
    nout = nL + nR
    mx = 0
    for t in Terms:
        mx = max(mx, t.Isite, t.Jsite)
    if Terms and mx >= nout:
        raise OracleError(1, "Maximum site index from Terms has to be less than the total number of sites")
    out = []
    for t in Terms:
        if 0 <= t.Isite < nL and nL <= t.Jsite < nout:
            if t.a == 0.0:
                continue
            out.append(t._replace(Jsite=nout - 1 - t.Jsite))
        elif 0 <= t.Isite < nL and 0 <= t.Jsite < nL:
            pass
        elif nL <= t.Isite < nout and nL <= t.Jsite < nout:
            pass
        else:
            raise OracleError(1, f"Invalid term: Isite={t.Isite} Jsite={t.Jsite}")
    return out


def _block_op(blk, op, isite):
    """GetBlockMat (src/DMRGKron.cpp:884-885); Sm = Sp^T (src/DMRGBlock.cpp:630-632)."""
    if op == OpSp:
        return blk.SpData[isite]
    if op == OpSz:
        return blk.SzData[isite]
    if op == OpSm:
        return _sorted(blk.SpData[isite].T)
    raise OracleError(1, "bad op")


def kron_sum_terms(kb, Terms):
    """Term list [H_L(x)1, 1(x)H_R, LR terms...] as (a, OpA, A, OpB, B) (src/DMRGKron.cpp:936-981)."""
    L, R = kb.LeftBlock, kb.RightBlock
    out = [(1.0, OpSz, _sorted(L.H), OpEye, None), (1.0, OpEye, None, OpSz, _sorted(R.H))]
    cacheL, cacheR = {}, {}
    for t in classify_terms(kb, Terms):
        keyL, keyR = (t.Iop, t.Isite), (t.Jop, t.Jsite)
        if keyL not in cacheL:
            cacheL[keyL] = _block_op(L, *keyL)
        if keyR not in cacheR:
            cacheR[keyR] = _block_op(R, *keyR)
        out.append((t.a, t.Iop, cacheL[keyL], t.Jop, cacheR[keyR]))
    return out


def _col_block(kb, IL, IR, opA, opB):
    """Column KronBlock reached from row block (IL,IR) by a term with left/right op types
    (src/DMRGKron.cpp:1754-1766): only the LEFT op type selects the KronBlock, the right one the width."""
    sA = 0 if opA in (OpEye, OpSz) else opA
    sB = 0 if opB in (OpEye, OpSz) else opB
    if sA == 0:
        kc = kb.Map(IL, IR)
    else:
        kc = kb.Map(IL + sA, IR - sA)
    return kc, sA, sB


def KronSumConstruct_explicit(kb, Terms, ks_tol=KS_TOL):
    """Explicit sparse H = sum_t a_t A_t (x) B_t restricted to the KronBlocks (KronSumFillMatrix,
    src/DMRGKron.cpp:1340-1477), assembled block-wise; entries with |v| < ks_tol are dropped (:1449-1454)."""
    Lq, Rq = kb.LeftBlock.Magnetization, kb.RightBlock.Magnetization
    N = kb.NumStates()
    acc = sp.csr_matrix((N, N), dtype=np.float64)
    for (a, opA, A, opB, B) in kron_sum_terms(kb, Terms):
        R, C, V = [], [], []
        for k, (_, IL, IR, size) in enumerate(kb.kb):
            kc, sA, sB = _col_block(kb, IL, IR, opA, opB)
            ILc, IRc = IL + sA, IR + sB
            # flg[SideLeft]/flg[SideRight] (:1434) and the fws lookup; a term whose shifted KronBlock is
            # absent contributes nothing (Offsets()==-1 rows are never touched for valid Sz-conserving terms)
            if not (0 <= ILc < Lq.NumSectors() and 0 <= IRc < Rq.NumSectors()) or kc < 0:
                continue
            if kb.kb[kc][1] != ILc or kb.kb[kc][2] != IRc:
                continue
            subA = (sp.identity(Lq.qn_size[IL], format="csr") if opA == OpEye else
                    A[Lq.qn_offset[IL]:Lq.qn_offset[IL + 1], Lq.qn_offset[ILc]:Lq.qn_offset[ILc + 1]])
            subB = (sp.identity(Rq.qn_size[IR], format="csr") if opB == OpEye else
                    B[Rq.qn_offset[IR]:Rq.qn_offset[IR + 1], Rq.qn_offset[IRc]:Rq.qn_offset[IRc + 1]])
            big = sp.kron(subA, subB, format="coo")
            R.append(big.row + kb.kb_offset[k])
            C.append(big.col + kb.kb_offset[kc])
            V.append(a * big.data)
        if R:
            acc = acc + sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(N, N)).tocsr()
    acc = acc.tocsr()
    acc.data[np.abs(acc.data) < ks_tol] = 0.0
    acc.eliminate_zeros()
    return _sorted(acc)


class KronSumOperator:
    """The same operator as KronSumConstruct_explicit, matrix-free: y_k = sum_t a_t A_t[IL, IL'] X_k' B_t[IR, IR']^T per KronBlock,
    with the left operators of the terms that share a right operator summed first (the map the reference builds at
    src/DMRGKron.cpp:955-960).  For superblocks whose explicit matrix would not fit the oracle's time budget (10^5 states and up:
    tests/golden/make_engine_golden_large_m.py); checked against the explicit matrix in tests/test_oracle_golden.py."""

    def __init__(self, kb, Terms):
        self.kb = kb
        Lq, Rq = kb.LeftBlock.Magnetization, kb.RightBlock.Magnetization
        N = kb.NumStates()
        self.shape, self.dtype = (N, N), np.dtype(np.float64)
        groups = {}
        for (a, opA, A, opB, B) in kron_sum_terms(kb, Terms):
            key = (opA if B is None else opB, id(A) if B is None else id(B), B is None)
            g = groups.setdefault(key, dict(opA=opA, opB=opB, A=None, B=B))
            assert g["opA"] == opA and g["opB"] == opB
            if A is not None:
                g["A"] = a * A if g["A"] is None else g["A"] + a * A
            else:
                g["scale"] = g.get("scale", 0.0) + a
        self.tasks = []                       # (row block k, column block kc, dense subA or None, scale, dense subB or None)
        for g in groups.values():
            for k, (_, IL, IR, size) in enumerate(kb.kb):
                kc, sA, sB = _col_block(kb, IL, IR, g["opA"], g["opB"])
                ILc, IRc = IL + sA, IR + sB
                if not (0 <= ILc < Lq.NumSectors() and 0 <= IRc < Rq.NumSectors()) or kc < 0:
                    continue
                if kb.kb[kc][1] != ILc or kb.kb[kc][2] != IRc:
                    continue
                subA = None if g["A"] is None else g["A"][Lq.qn_offset[IL]:Lq.qn_offset[IL + 1], Lq.qn_offset[ILc]:Lq.qn_offset[ILc + 1]]
                subB = None if g["B"] is None else g["B"][Rq.qn_offset[IR]:Rq.qn_offset[IR + 1], Rq.qn_offset[IRc]:Rq.qn_offset[IRc + 1]]
                if (subA is not None and subA.nnz == 0) or (subB is not None and subB.nnz == 0):
                    continue
                self.tasks.append((k, kc, None if subA is None else np.ascontiguousarray(subA.toarray()), g.get("scale", 1.0),
                                   None if subB is None else np.ascontiguousarray(subB.toarray().T)))
        self.dims = [(Lq.qn_size[IL], Rq.qn_size[IR]) for (_, IL, IR, _) in kb.kb]

    def matvec(self, x):
        kb = self.kb
        x = np.asarray(x, dtype=np.float64).ravel()
        X = [x[kb.kb_offset[k]:kb.kb_offset[k + 1]].reshape(self.dims[k]) for k in range(kb.size())]
        y = np.zeros_like(x)
        Y = [y[kb.kb_offset[k]:kb.kb_offset[k + 1]].reshape(self.dims[k]) for k in range(kb.size())]
        for (k, kc, subA, scale, subBT) in self.tasks:
            T = X[kc] if subBT is None else X[kc] @ subBT
            Y[k] += scale * T if subA is None else subA @ T
        return y

    def as_linear_operator(self):
        import scipy.sparse.linalg as spla
        return spla.LinearOperator(self.shape, matvec=self.matvec, dtype=self.dtype)


# --------------------------------------------------------------------------------------------------
#  Shell path: per-(row, term) descriptors and the literal matvec
# --------------------------------------------------------------------------------------------------
class ShellCtx:
    """Flat-array form of KronSumShellCtx / KronSumTermRow (include/DMRGKron.hpp:85-112).

    For every term t: CSR of A_t and B_t (identity encoded as a 1-entry-per-row CSR whose column is the
    row itself, src/DMRGKron.cpp:1781-1797).  For every (row, term): bks_L, col_NStatesR, fws_O, valid.
    """

    def __init__(self, kb, Terms):
        self.kb = kb
        Lq, Rq = kb.LeftBlock.Magnetization, kb.RightBlock.Magnetization
        terms = kron_sum_terms(kb, Terms)
        self.nterms = len(terms)
        self.term_a = np.array([t[0] for t in terms], dtype=np.float64)
        ks, ILs, IRs, _, _, gL, gR = kb.rows()
        self.N = kb.NumStates()
        self.Rows_L, self.Rows_R = gL.astype(np.int64), gR.astype(np.int64)
        nsL, nsR = Lq.NumStates(), Rq.NumStates()
        eyeL = sp.identity(nsL, format="csr", dtype=np.float64)
        eyeR = sp.identity(nsR, format="csr", dtype=np.float64)
        self.A = [eyeL if t[1] == OpEye else t[2] for t in terms]
        self.B = [eyeR if t[3] == OpEye else t[4] for t in terms]
        nb = kb.size()
        # per (KronBlock, term) constants, broadcast to rows below
        bksL = np.zeros((nb, self.nterms), dtype=np.int64)
        colNR = np.zeros((nb, self.nterms), dtype=np.int64)
        fwsO = np.zeros((nb, self.nterms), dtype=np.int64)
        valid = np.zeros((nb, self.nterms), dtype=np.int64)
        for k, (_, IL, IR, _) in enumerate(kb.kb):
            fws_LOP = {OpEye: kb.kb_offset[k], OpSz: kb.kb_offset[k],
                       OpSp: kb.Offsets(IL + 1, IR - 1), OpSm: kb.Offsets(IL - 1, IR + 1)}  # :1754-1759
            nR_ROP = {OpEye: Rq.qn_size[IR], OpSz: Rq.qn_size[IR],
                      OpSp: Rq.Sizes(IR + 1), OpSm: Rq.Sizes(IR - 1)}  # :1761-1766
            for it, (a, opA, A, opB, B) in enumerate(terms):
                bL, fL = Lq.OpBlockToGlobalRangeStart(IL, OpSz if opA == OpEye else opA)  # :1780,1785
                bR, fR = Rq.OpBlockToGlobalRangeStart(IR, OpSz if opB == OpEye else opB)  # :1791,1796
                if not (fL and fR):  # :1799-1801
                    continue
                if fws_LOP[opA] == -1 or nR_ROP[opB] == -1:
                    # the reference would index x with a garbage offset here; valid Sz-conserving term lists
                    # never reach it because the operator rows are then empty (nz_L*nz_R==0)
                    continue
                bksL[k, it], colNR[k, it] = bL, nR_ROP[opB]
                fwsO[k, it] = fws_LOP[opA] - bR  # :1803
                valid[k, it] = 1
        self.blk_of_row = ks.astype(np.int64)
        self.bks_L, self.col_NStatesR, self.fws_O, self.valid = bksL, colNR, fwsO, valid

    def apply_literal(self, x):
        """Pure-python statement of src/DMRGKron.cpp:1842-1864 (tiny sizes only)."""
        y = np.zeros(self.N)
        for ir in range(self.N):
            k = self.blk_of_row[ir]
            yval = 0.0
            for it in range(self.nterms):
                if not self.valid[k, it]:
                    continue
                A, B = self.A[it], self.B[it]
                rl, rr = self.Rows_L[ir], self.Rows_R[ir]
                for l in range(A.indptr[rl], A.indptr[rl + 1]):
                    idx = (A.indices[l] - self.bks_L[k, it]) * self.col_NStatesR[k, it] + self.fws_O[k, it]
                    temp = self.term_a[it] * A.data[l]
                    for r in range(B.indptr[rr], B.indptr[rr + 1]):
                        yval += temp * B.data[r] * x[idx + B.indices[r]]
            y[ir] = yval
        return y
