/** @file QuantumNumbers.hpp
    Sz-sector bookkeeping of a block basis: descending sector list, sizes, prefix offsets, and the translation of an
    operator's sector shift into a column range.  Same public interface and error behaviour as the reference class
    (reference include/QuantumNumbers.hpp:30-239, src/QuantumNumbers.cpp:9-201); integer-only host code that feeds
    the device kernels their sector tables. */
#ifndef DMRGX_QUANTUM_NUMBERS_HPP
#define DMRGX_QUANTUM_NUMBERS_HPP

#include <vector>
#include <cassert>
#include "petsc_compat.hpp"

class QuantumNumbers
{
public:
    PetscErrorCode Initialize(const MPI_Comm& comm_in, const std::vector<PetscReal>& list_in, const std::vector<PetscInt>& size_in)
    {
        if (list_in.empty()) SETERRQ(comm_in, PETSC_ERR_ARG_WRONG, "Initialization error: Empty input list.");
        if (list_in.size() != size_in.size()) SETERRQ(comm_in, PETSC_ERR_ARG_WRONG, "Initialization error: Input list sizes mismatch.");
        for (size_t i = 1; i < list_in.size(); ++i)
            if (list_in[i] >= list_in[i - 1]) SETERRQ(PETSC_COMM_SELF, 1, "qn_list_in must be sorted descending.");
        mpi_comm = comm_in;
        qn_list = list_in; qn_size = size_in;
        num_sectors = (PetscInt)qn_list.size();
        qn_offset.assign(num_sectors + 1, 0);
        for (PetscInt i = 0; i < num_sectors; ++i) qn_offset[i + 1] = qn_offset[i] + qn_size[i];
        num_states = qn_offset.back();
        initialized = PETSC_TRUE;
        return 0;
    }
    PetscErrorCode CheckInitialized() const
    {
        if (PetscUnlikely(!initialized)) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "QuantumNumbers object not yet initialized.");
        return 0;
    }
    PetscBool Initialized() const { return initialized; }
    MPI_Comm MPIComm() const { return mpi_comm; }
    PetscInt NumSectors() const { assert(initialized); return num_sectors; }
    PetscInt NumStates() const { assert(initialized); return num_states; }
    std::vector<PetscReal> List() const { assert(initialized); return qn_list; }
    const std::vector<PetscReal>& ListRef() const { assert(initialized); return qn_list; }
    PetscReal List(const PetscInt& idx) const { assert(initialized); return (0 <= idx && idx < num_sectors) ? qn_list[idx] : -1; }
    std::vector<PetscInt> Offsets() const { assert(initialized); return qn_offset; }
    PetscInt Offsets(const PetscInt& idx) const { assert(initialized); return (0 <= idx && idx < num_sectors) ? qn_offset[idx] : -1; }
    std::vector<PetscInt> Sizes() const { assert(initialized); return qn_size; }
    PetscInt Sizes(const PetscInt& idx) const { assert(initialized); return (0 <= idx && idx < num_sectors) ? qn_size[idx] : -1; }
    /** sector sizes as the int32 table the device ABI takes */
    std::vector<int32_t> Sizes32() const { return std::vector<int32_t>(qn_size.begin(), qn_size.end()); }

    PetscErrorCode BlockIdxToGlobalRange(const PetscInt& BlockIdx, PetscInt& GlobIdxStart, PetscInt& GlobIdxEnd) const
    {
        PetscErrorCode ierr = CheckInitialized(); CHKERRQ(ierr);
        if (PetscUnlikely(BlockIdx < 0 || BlockIdx >= num_sectors))
            SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Given BlockIdx (%lld) out of bounds [0, %lld).", LLD(BlockIdx), LLD(num_sectors));
        GlobIdxStart = qn_offset[BlockIdx]; GlobIdxEnd = qn_offset[BlockIdx + 1];
        return 0;
    }
    /** Column range of sector BlockIdx+BlockShift; flg = PETSC_FALSE when that sector does not exist. */
    PetscErrorCode OpBlockToGlobalRange(const PetscInt& BlockIdx, const PetscInt& BlockShift, PetscInt& GlobIdxStart, PetscInt& GlobIdxEnd, PetscBool& flg) const
    {
        PetscErrorCode ierr = CheckInitialized(); CHKERRQ(ierr);
        if (PetscUnlikely(BlockIdx < 0 || BlockIdx >= num_sectors))
            SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Given BlockIdx (%lld) out of bounds [0, %lld).", LLD(BlockIdx), LLD(num_sectors));
        const PetscInt out = BlockIdx + BlockShift;
        if (out < 0 || out >= num_sectors) { flg = PETSC_FALSE; return 0; }
        flg = PETSC_TRUE;
        GlobIdxStart = qn_offset[out]; GlobIdxEnd = qn_offset[out + 1];
        return 0;
    }
    PetscInt OpBlockToGlobalRangeStart(const PetscInt& BlockIdx, const PetscInt& BlockShift, PetscBool& flg) const
    {
        PetscInt s = 0, e = 0;
        PetscErrorCode ierr = OpBlockToGlobalRange(BlockIdx, BlockShift, s, e, flg);
        assert(!ierr); (void)ierr;
        return s;
    }
    PetscErrorCode QNToGlobalRange(const PetscReal& QNValue, PetscInt& GlobIdxStart, PetscInt& GlobIdxEnd) const
    {
        PetscErrorCode ierr = CheckInitialized(); CHKERRQ(ierr);
        for (PetscInt b = 0; b < num_sectors; ++b)
            if (qn_list[b] == QNValue) { GlobIdxStart = qn_offset[b]; GlobIdxEnd = qn_offset[b + 1]; return 0; }
        SETERRQ1(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Given QNValue (%g) not found.", QNValue);
    }
    PetscErrorCode GlobalIdxToBlockIdx(const PetscInt& GlobIdx, PetscInt& BlockIdx) const
    {
        PetscErrorCode ierr = CheckInitialized(); CHKERRQ(ierr);
        if (PetscUnlikely(GlobIdx < 0 || GlobIdx >= num_states))
            SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Given GlobIdx (%lld) out of bounds [0, %lld).", LLD(GlobIdx), LLD(num_states));
        BlockIdx = -1;
        while (GlobIdx >= qn_offset[BlockIdx + 1]) ++BlockIdx;
        return 0;
    }
    PetscErrorCode GlobalIdxToBlockIdx(const PetscInt& GlobIdx, PetscInt& BlockIdx, PetscInt& LocIdx) const
    {
        PetscErrorCode ierr = GlobalIdxToBlockIdx(GlobIdx, BlockIdx); CHKERRQ(ierr);
        LocIdx = GlobIdx - qn_offset[BlockIdx];
        return 0;
    }
    PetscErrorCode GlobalIdxToQN(const PetscInt& GlobIdx, PetscReal& QNValue) const
    {
        PetscInt b; PetscErrorCode ierr = GlobalIdxToBlockIdx(GlobIdx, b); CHKERRQ(ierr);
        QNValue = qn_list[b];
        return 0;
    }
    PetscErrorCode BlockIdxToGlobalIdx(const PetscInt& BlockIdx, const PetscInt& LocIdx, PetscInt& GlobIdx) const
    {
        PetscErrorCode ierr = CheckInitialized(); CHKERRQ(ierr);
        if (PetscUnlikely(BlockIdx < 0 || BlockIdx >= num_sectors))
            SETERRQ2(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "Given BlockIdx (%lld) out of bounds [0, %lld).", LLD(BlockIdx), LLD(num_sectors));
        GlobIdx = qn_offset[BlockIdx] + LocIdx;
        return 0;
    }
    PetscInt BlockIdxToGlobalIdx(const PetscInt& BlockIdx, const PetscInt& LocIdx) const
    {
        assert(initialized && 0 <= BlockIdx && BlockIdx < num_sectors);
        return qn_offset[BlockIdx] + LocIdx;
    }
    PetscErrorCode PrintQNs()
    {
        printf("[ ");
        for (PetscInt i = 0; i < num_sectors; ++i) for (PetscInt j = 0; j < qn_size[i]; ++j) printf("%g ", qn_list[i]);
        printf(" ]\n");
        return 0;
    }
private:
    MPI_Comm mpi_comm = PETSC_COMM_SELF;
    PetscInt num_sectors = 0, num_states = 0;
    std::vector<PetscReal> qn_list;
    std::vector<PetscInt> qn_offset, qn_size;
    PetscBool initialized = PETSC_FALSE;
};

/** Walks a range of basis states keeping track of the sector each one belongs to. */
class QuantumNumbersIterator
{
public:
    explicit QuantumNumbersIterator(const QuantumNumbers& QN_in) : QN(QN_in), iend_(QN_in.NumStates()) {}
    QuantumNumbersIterator(const QuantumNumbers& QN_in, const PetscInt& GlobIdxStart, const PetscInt& GlobIdxEnd)
        : QN(QN_in), istart_(GlobIdxStart), iend_(GlobIdxEnd), idx_(GlobIdxStart)
    {
        if (istart_ != iend_) { PetscErrorCode ierr = QN.GlobalIdxToBlockIdx(istart_, blockidx_); assert(!ierr); (void)ierr; }
    }
    PetscInt Idx() const { return idx_; }
    PetscInt BlockIdx() const { return blockidx_; }
    PetscInt IdxStart() const { return istart_; }
    PetscInt IdxEnd() const { return iend_; }
    PetscInt LocIdx() const { return idx_ - QN.Offsets()[blockidx_]; }
    bool Loop() const { return idx_ < iend_; }
    PetscInt Steps() const { return idx_ - istart_; }
    QuantumNumbersIterator& operator++()
    {
        ++idx_;
        if (idx_ < QN.NumStates() && idx_ >= QN.Offsets()[blockidx_ + 1]) ++blockidx_;
        return *this;
    }
    PetscErrorCode OpBlockToGlobalRange(const PetscInt& BlockShift, PetscInt& GlobIdxStart, PetscInt& GlobIdxEnd, PetscBool& flg) const
    {
        return QN.OpBlockToGlobalRange(blockidx_, BlockShift, GlobIdxStart, GlobIdxEnd, flg);
    }
private:
    const QuantumNumbers& QN;
    PetscInt istart_ = 0, iend_ = 0, idx_ = 0, blockidx_ = 0;
};

#endif
