"""Oracle: Block::SpinBase restated with scipy CSR operators.

Follows src/DMRGBlock.cpp:31-211 (initialisers), :375-620 (validity checks), :623-652 (Sm), :677-823
(RotateOperators), :1106-1225 (single-site operators).  Disk spill (:835-1103) is out of scope.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import scipy.sparse as sp

from .qn import (QuantumNumbers, OracleError, OpSm, OpSz, OpSp,
                 PETSC_ERR_ARG_OUTOFRANGE, PETSC_ERR_ARG_WRONG, PETSC_ERR_ARG_CORRUPT)


def csr_from_rows(n, rows):
    """Build an n x n CSR keeping explicit (structural) zeros: rows = {row: [(col, val), ...]}.

    PETSc's MatSetValues keeps a zero-valued entry as a structural non-zero and the reference's
    known-answer tables check those (tests/UnitTests_Misc.cpp:23-66), so the oracle keeps them too.
    """
    indptr, indices, data = [0], [], []
    for r in range(n):
        ent = sorted(rows.get(r, []))
        indices += [c for c, _ in ent]
        data += [float(v) for _, v in ent]
        indptr.append(len(indices))
    m = sp.csr_matrix((np.array(data, dtype=np.float64), np.array(indices, dtype=np.int64),
                       np.array(indptr, dtype=np.int64)), shape=(n, n))
    m.has_sorted_indices = True
    return m


class Block:
    """Spin block: per-site Sz(i), Sp(i), block Hamiltonian H, Magnetization sectors."""

    loc_dim = 2
    loc_qn_list = [+0.5, -0.5]  # src/DMRGBlock.cpp:81-85
    loc_qn_size = [1, 1]

    def __init__(self):
        self.init = False
        self.num_sites = 0
        self.num_states = 0
        self.Magnetization = None
        self.SzData, self.SpData, self.SmData = [], [], []
        self.H = None
        self.init_Sm = False

    # -- initialisers (src/DMRGBlock.cpp:44-211) -------------------------------------------------
    @classmethod
    def single_site(cls):
        """Initialize(comm, 1, PETSC_DEFAULT): spin-1/2 site (src/DMRGBlock.cpp:123-137,1131-1136,1193-1195)."""
        b = cls()
        b.num_sites, b.num_states = 1, cls.loc_dim
        b.SzData = [csr_from_rows(2, {0: [(0, +0.5)], 1: [(1, -0.5)]})]
        b.SpData = [csr_from_rows(2, {0: [(1, +1.0)]})]
        b.SmData = [None]
        b.H = csr_from_rows(2, {})
        b.Magnetization = QuantumNumbers(cls.loc_qn_list, cls.loc_qn_size)
        b.init = True
        return b

    @classmethod
    def with_sectors(cls, num_sites, qn_list, qn_size):
        """Initialize(comm, nsites, qn_list, qn_size): empty operators of the right size (:140-147,174-197)."""
        b = cls()
        b.Magnetization = QuantumNumbers(qn_list, qn_size)
        b.num_sites, b.num_states = int(num_sites), b.Magnetization.NumStates()
        n = b.num_states
        b.SzData = [csr_from_rows(n, {}) for _ in range(b.num_sites)]
        b.SpData = [csr_from_rows(n, {}) for _ in range(b.num_sites)]
        b.SmData = [None] * b.num_sites
        b.H = None
        b.init = True
        return b

    def NumSites(self):
        return self.num_sites

    def NumStates(self):
        return self.num_states

    def Sz(self, i):
        return self.SzData[i]

    def Sp(self, i):
        return self.SpData[i]

    def Sm(self, i):
        if not self.init_Sm:
            raise RuntimeError("Sm not initialised")  # include/DMRGBlock.hpp:353-369
        return self.SmData[i]

    # -- checks (src/DMRGBlock.cpp:375-620) ------------------------------------------------------
    def CheckOperatorArray(self, op_type):
        arr = {OpSm: self.SmData, OpSz: self.SzData, OpSp: self.SpData}.get(op_type)
        if arr is None:
            raise OracleError(PETSC_ERR_ARG_WRONG, "Incorrect operator type.")
        for i in range(self.num_sites):
            if arr[i] is None:
                raise OracleError(PETSC_ERR_ARG_CORRUPT, f"[{i}] matrix not yet created.")
            M, N = arr[i].shape
            if M != N:
                raise OracleError(PETSC_ERR_ARG_WRONG, f"[{i}] matrix not square.")
            if M != self.num_states:
                raise OracleError(PETSC_ERR_ARG_WRONG, f"[{i}] matrix dimension does not match the number of states.")

    def CheckOperators(self):
        if not self.init:
            raise OracleError(PETSC_ERR_ARG_CORRUPT, "Block not yet initialized")
        self.CheckOperatorArray(OpSz)
        self.CheckOperatorArray(OpSp)
        if self.init_Sm:
            self.CheckOperatorArray(OpSm)

    def CheckSectors(self):
        if self.num_states != self.Magnetization.NumStates():
            raise OracleError(PETSC_ERR_ARG_WRONG, "The number of states in the Magnetization object "
                              "and the internal value do not match.")

    def MatCheckOperatorBlocks(self, op_type, mat):
        """Every row's first/last column must lie in the column range of sector+shift (:520-598).

        Like the reference, only the first and the last entry of each (sorted) row are inspected and the
        out-of-range case raises PETSC_ERR_ARG_OUTOFRANGE through CheckIndex (:27-29).
        """
        self.CheckSectors()
        qn = self.Magnetization
        if mat.shape[0] != qn.NumStates():
            raise OracleError(1, "Incorrect number of rows.")
        indptr, indices = mat.indptr, mat.indices
        for blk in range(qn.NumSectors()):
            cs, ce, flg = qn.OpBlockToGlobalRange(blk, op_type)
            for row in range(qn.qn_offset[blk], qn.qn_offset[blk + 1]):
                a, b = indptr[row], indptr[row + 1]
                if a == b:
                    continue
                if not flg:
                    cs = ce = 0  # the reference leaves the range stale; any entry is then out of bounds
                for col in (indices[a], indices[b - 1]):
                    if col < cs or col >= ce:
                        raise OracleError(PETSC_ERR_ARG_OUTOFRANGE,
                                          f"On row {row}, index {col} out of bounds [{cs},{ce})")

    def MatOpCheckOperatorBlocks(self, op_type, isite):
        if isite >= self.num_sites:
            raise OracleError(PETSC_ERR_ARG_OUTOFRANGE, f"Input isite ({isite}) out of bounds")
        arr = {OpSm: self.SmData, OpSz: self.SzData, OpSp: self.SpData}[op_type]
        self.MatCheckOperatorBlocks(op_type, arr[isite])

    def CheckOperatorBlocks(self):
        self.CheckOperators()
        for i in range(self.num_sites):
            self.MatOpCheckOperatorBlocks(OpSz, i)
        for i in range(self.num_sites):
            self.MatOpCheckOperatorBlocks(OpSp, i)

    # -- Sm (src/DMRGBlock.cpp:623-652) ----------------------------------------------------------
    def CreateSm(self):
        if self.init_Sm:
            raise OracleError(1, "Sm was previously initialized. Call DestroySm() first.")
        self.SmData = [self.SpData[i].T.tocsr() for i in range(self.num_sites)]
        for m in self.SmData:
            m.sort_indices()
        self.init_Sm = True

    def DestroySm(self):
        self.SmData = [None] * self.num_sites
        self.init_Sm = False

    # -- rotation (src/DMRGBlock.cpp:677-823) ----------------------------------------------------
    def RotateOperators(self, source, RotMatT):
        """Sp'(i)=RotMatT.Sp(i).RotMat, same for Sz(i) and H (:763-772); RotMatT is (m x 2m) CSR."""
        nr, nc = RotMatT.shape
        if nc != source.NumStates() or nr != self.num_states or source.NumSites() != self.num_sites:
            raise OracleError(1, "RotMatT_in incorrect shape")
        RotMat = RotMatT.T.tocsr()
        self.SpData = [_sorted(RotMatT @ source.SpData[i] @ RotMat) for i in range(self.num_sites)]
        self.SzData = [_sorted(RotMatT @ source.SzData[i] @ RotMat) for i in range(self.num_sites)]
        self.H = _sorted(RotMatT @ source.H @ RotMat)
        self.SmData = [None] * self.num_sites
        self.init_Sm = False
        self.CheckOperatorBlocks()  # :814


def _sorted(m):
    m = m.tocsr()
    m.sort_indices()
    return m
