"""Oracle: sector bookkeeping.  Restates src/QuantumNumbers.cpp:9-201 and include/QuantumNumbers.hpp:60-237.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

# include/DMRGBlock.hpp:19-27 -- the operator type doubles as the sector-index shift of the column block.
OpSm, OpSz, OpSp, OpEye = -1, 0, +1, +2

# PETSc error codes the reference raises on this path (petscerror.h, PETSc 3.8): used so that the tests
# can assert the *same* code the reference's own tests assert (tests/UnitTests_DMRGBlock.cpp:120).
PETSC_ERR_ARG_OUTOFRANGE = 63
PETSC_ERR_ARG_WRONG = 62
PETSC_ERR_ARG_WRONGSTATE = 73
PETSC_ERR_ARG_CORRUPT = 64


class OracleError(Exception):
    """Carries the PetscErrorCode the reference would have returned."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class QuantumNumbers:
    """Sorted-descending Sz sector list with sizes and prefix offsets (src/QuantumNumbers.cpp:9-50)."""

    def __init__(self, qn_list, qn_size):
        qn_list = [float(q) for q in qn_list]
        qn_size = [int(s) for s in qn_size]
        if len(qn_list) == 0:
            raise OracleError(PETSC_ERR_ARG_WRONG, "Initialization error: Empty input list.")
        if len(qn_list) != len(qn_size):
            raise OracleError(PETSC_ERR_ARG_WRONG, "Initialization error: Input list sizes mismatch.")
        for a, b in zip(qn_list[:-1], qn_list[1:]):  # :31-39 strictly descending
            if b >= a:
                raise OracleError(1, "qn_list_in must be sorted descending.")
        self.qn_list = qn_list
        self.qn_size = qn_size
        self.qn_offset = [0]
        for s in qn_size:  # :43-46
            self.qn_offset.append(self.qn_offset[-1] + s)
        self.num_sectors = len(qn_list)
        self.num_states = self.qn_offset[-1]

    # include/QuantumNumbers.hpp:100-141 -- out-of-range accessors return -1
    def List(self, idx=None):
        if idx is None:
            return list(self.qn_list)
        return self.qn_list[idx] if 0 <= idx < self.num_sectors else -1

    def Sizes(self, idx=None):
        if idx is None:
            return list(self.qn_size)
        return self.qn_size[idx] if 0 <= idx < self.num_sectors else -1

    def Offsets(self, idx=None):
        if idx is None:
            return list(self.qn_offset)
        return self.qn_offset[idx] if 0 <= idx < self.num_sectors else -1

    def NumStates(self):
        return self.num_states

    def NumSectors(self):
        return self.num_sectors

    def OpBlockToGlobalRange(self, block_idx, shift):
        """src/QuantumNumbers.cpp:72-96 -> (start, end, flg)."""
        if block_idx < 0 or block_idx >= self.num_sectors:
            raise OracleError(PETSC_ERR_ARG_OUTOFRANGE, f"Given BlockIdx ({block_idx}) out of bounds")
        out = block_idx + shift
        if out < 0 or out >= self.num_sectors:
            return 0, 0, False
        return self.qn_offset[out], self.qn_offset[out + 1], True

    def OpBlockToGlobalRangeStart(self, block_idx, shift):
        """include/QuantumNumbers.hpp:169-180 -> (start, flg)."""
        s, _, flg = self.OpBlockToGlobalRange(block_idx, shift)
        return s, flg

    def GlobalIdxToBlockIdx(self, glob_idx):
        """src/QuantumNumbers.cpp:122-154 -> (block, local)."""
        if glob_idx < 0 or glob_idx >= self.num_states:
            raise OracleError(PETSC_ERR_ARG_OUTOFRANGE, f"Given GlobIdx ({glob_idx}) out of bounds")
        blk = -1
        while glob_idx >= self.qn_offset[blk + 1]:
            blk += 1
        return blk, glob_idx - self.qn_offset[blk]

    def BlockIdxToGlobalIdx(self, blk, loc):
        """src/QuantumNumbers.cpp:193-201."""
        return self.qn_offset[blk] + loc

    def sector_of_rows(self):
        """Vector: sector index of every basis state (helper for vectorised checks)."""
        return np.repeat(np.arange(self.num_sectors), self.qn_size)
