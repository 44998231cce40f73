/** @file DMRGKron.hpp
    Sector-pair ("KronBlock") algebra of two blocks: the ordered pair list and its offsets (KronBlocks_t), block
    enlargement (KronEye_Explicit) and the superblock Hamiltonian as a matrix-free device plan (KronSumConstruct ->
    MatMult_KronSumShell).  Public interface follows reference include/DMRGKron.hpp:22-663 / src/DMRGKron.cpp; the
    implementation is new:
      - enlargement is metadata only for the site operators: O (x) 1 and 1 (x) s become cell views / scaled-identity
        cells over the SAME device buffers (no copy, no flops); only the enlarged block Hamiltonian is assembled, by
        one batch of dense-cell accumulates on the device (dmrgx_cells_axpy);
      - the superblock matrix is always matrix-free: KronSumConstruct fills a dmrgx_kron_desc and creates the HIP plan
        (grouped MFMA GEMMs), MatMult_KronSumShell applies it.  The explicit MPIAIJ superblock matrix of the reference
        (-do_shell 0) is not built. */
#ifndef DMRGX_DMRGKRON_HPP
#define DMRGX_DMRGKRON_HPP

#include <algorithm>
#include <map>
#include <set>
#include <tuple>
#include <vector>
#include "DMRGBlock.hpp"
#include "Hamiltonians.hpp"

/** (quantum number, left sector index, right sector index, number of states) */
typedef std::tuple<PetscReal, PetscInt, PetscInt, PetscInt> KronBlock_t;

class KronBlocksIterator;

class KronBlocks_t
{
    friend class KronBlocksIterator;
public:
    KronBlocks_t(Block::SpinBase& LeftBlock, Block::SpinBase& RightBlock, const std::vector<PetscReal>& QNSectors, FILE* fp_prealloc, const PetscInt& GlobIdx)
        : GlobIdx(GlobIdx), LeftBlock(LeftBlock), RightBlock(RightBlock), fp_prealloc(fp_prealloc)
    {
        if (!LeftBlock.Initialized()) throw std::runtime_error("Left input block not initialized.");
        if (!RightBlock.Initialized()) throw std::runtime_error("Right input block not initialized.");
        mpi_comm = LeftBlock.MPIComm();
        if (mpi_comm != RightBlock.MPIComm()) throw std::runtime_error("Left and right blocks must have the same communicator.");
        const std::vector<PetscReal> ql = LeftBlock.Magnetization.List(), qr = RightBlock.Magnetization.List();
        const std::vector<PetscInt> sl = LeftBlock.Magnetization.Sizes(), sr = RightBlock.Magnetization.Sizes();
        const std::set<PetscReal> want(QNSectors.begin(), QNSectors.end());
        /* nested IL-then-IR enumeration; quantum numbers are halves, compared exactly */
        for (size_t IL = 0; IL < ql.size(); ++IL)
            for (size_t IR = 0; IR < qr.size(); ++IR) {
                const PetscReal qn = ql[IL] + qr[IR];
                if (want.empty() || want.count(qn)) KronBlocks.push_back(std::make_tuple(qn, (PetscInt)IL, (PetscInt)IR, sl[IL] * sr[IR]));
            }
        if (want.empty()) std::stable_sort(KronBlocks.begin(), KronBlocks.end(), [](const KronBlock_t& a, const KronBlock_t& b) { return std::get<0>(a) > std::get<0>(b); });
        num_blocks = (PetscInt)KronBlocks.size();
        PetscInt sum = 0, idx = 0;
        for (const KronBlock_t& kb : KronBlocks) {
            kb_list.push_back(std::get<0>(kb)); kb_size.push_back(std::get<3>(kb));
            kb_map[std::make_tuple(std::get<1>(kb), std::get<2>(kb))] = idx++;
            kb_offset.push_back(sum); sum += std::get<3>(kb);
        }
        kb_offset.push_back(sum);
        num_states = sum;
    }

    PetscInt size() const { return (PetscInt)KronBlocks.size(); }
    const std::vector<KronBlock_t>& data() const { return KronBlocks; }
    KronBlock_t data(size_t idx) const { return KronBlocks[idx]; }
    KronBlock_t operator[](size_t idx) const { return KronBlocks[idx]; }
    std::vector<PetscReal> List() const { return kb_list; }
    std::vector<PetscInt> Offsets() const { return kb_offset; }
    PetscInt Offsets(const PetscInt& idx) const { return kb_offset[idx]; }
    PetscReal QN(const PetscInt& idx) const { return std::get<0>(KronBlocks[idx]); }
    PetscInt LeftIdx(const PetscInt& idx) const { return std::get<1>(KronBlocks[idx]); }
    PetscInt RightIdx(const PetscInt& idx) const { return std::get<2>(KronBlocks[idx]); }
    PetscInt Sizes(const PetscInt& idx) const { return std::get<3>(KronBlocks[idx]); }
    std::vector<PetscInt> Sizes() const { return kb_size; }
    const Block::SpinBase& LeftBlockRef() const { return LeftBlock; }
    const Block::SpinBase& RightBlockRef() const { return RightBlock; }
    Block::SpinBase& LeftBlockRefMod() { return LeftBlock; }
    Block::SpinBase& RightBlockRefMod() { return RightBlock; }
    PetscInt Map(const PetscInt& lidx, const PetscInt& ridx) const
    {
        auto it = kb_map.find(std::make_tuple(lidx, ridx));
        return it == kb_map.end() ? -1 : it->second;
    }
    PetscInt Offsets(const PetscInt& lidx, const PetscInt& ridx) const { const PetscInt i = Map(lidx, ridx); return i >= 0 ? kb_offset[i] : -1; }
    PetscInt NumStates() const { return num_states; }

    PetscErrorCode KronSumSetShellMatrix(const PetscBool& do_shell_in)
    {
        if (!do_shell_in) SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_SUP, "KronSumSetShellMatrix: only the matrix-free (shell) superblock Hamiltonian is implemented.");
        do_shell = do_shell_in; return 0;
    }
    PetscErrorCode KronSumSetRedistribute(const PetscBool& in = PETSC_TRUE) { do_redistribute = in; return 0; }
    PetscErrorCode KronSumSetToleranceFromOptions() { return PetscOptionsGetReal(NULL, NULL, "-ks_tol", &ks_tol, NULL); }

    /** Keep the inter-block terms with a != 0 and renumber the right block's sites from the interface
        (L0 L1 .. R2 R1 R0): the filtering/reflection of the reference's KronSumConstruct. */
    PetscErrorCode ClassifyTerms(const std::vector<Hamiltonians::Term>& Terms, std::vector<Hamiltonians::Term>& TermsLR) const
    {
        const PetscInt nl = LeftBlock.NumSites(), nout = nl + RightBlock.NumSites();
        PetscInt mx = 0;
        for (const Hamiltonians::Term& t : Terms) mx = std::max(mx, std::max(t.Isite, t.Jsite));
        if (!Terms.empty() && mx >= nout)
            SETERRQ2(mpi_comm, 1, "Maximum site index from Terms (%lld) has to be less than the total number of sites in the blocks (%lld).", LLD(mx), LLD(nout));
        TermsLR.clear();
        for (const Hamiltonians::Term& t : Terms) {
            const bool iL = 0 <= t.Isite && t.Isite < nl, jL = 0 <= t.Jsite && t.Jsite < nl;
            const bool iR = nl <= t.Isite && t.Isite < nout, jR = nl <= t.Jsite && t.Jsite < nout;
            if (iL && jR) { if (t.a == PetscScalar(0.0)) continue; Hamiltonians::Term r = t; r.Jsite = nout - 1 - t.Jsite; TermsLR.push_back(r); }
            else if ((iL && jL) || (iR && jR)) {}
            else SETERRQ4(mpi_comm, 1, "Invalid term: Isite=%lld Jsite=%lld for nsites_left=%lld and nsites_right=%lld.", LLD(t.Isite), LLD(t.Jsite), LLD(nl), LLD(nout - nl));
        }
        return 0;
    }

    /** Superblock Hamiltonian restricted to this object's KronBlocks as a matrix-free device plan:
        MatOut applies  H_L (x) 1 + 1 (x) H_R + sum_t a_t A_t (x) B_t.  Release with MatDestroy_KronSumShell. */
    PetscErrorCode KronSumConstruct(const std::vector<Hamiltonians::Term>& Terms, Mat& MatOut)
    {
        PetscErrorCode ierr;
        ierr = LeftBlock.CheckOperators(); CHKERRQ(ierr);  ierr = LeftBlock.CheckSectors(); CHKERRQ(ierr);
        ierr = RightBlock.CheckOperators(); CHKERRQ(ierr); ierr = RightBlock.CheckSectors(); CHKERRQ(ierr);
        std::vector<Hamiltonians::Term> TermsLR;
        ierr = ClassifyTerms(Terms, TermsLR); CHKERRQ(ierr);
        /* the Hamiltonian plan is striped over the ranks of the communicator; the one-term correlator plans
           below stay whole on every rank (they act on the replicated ground state) */
        return BuildPlan(LeftBlock.H, RightBlock.H, TermsLR, MatOut, true);
    }

    /** Single product Mat_L (x) Mat_R on the KronBlocks (correlators), matrix-free like the Hamiltonian
        (src/DMRGKron.cpp:635-693 of the reference).  The operator types are checked against the matrices' sector
        blocks as the reference does; the matrices carry their own sector shift (a transposed view such as Sm(i) is read
        transposed, never materialised). */
    PetscErrorCode KronConstruct(const Mat& Mat_L, const Op_t& OpType_L, const Mat& Mat_R, const Op_t& OpType_R, Mat& MatOut)
    {
        PetscErrorCode ierr;
        ierr = LeftBlock.MatCheckOperatorBlocks(OpType_L, Mat_L); CHKERRQ(ierr);
        ierr = RightBlock.MatCheckOperatorBlocks(OpType_R, Mat_R); CHKERRQ(ierr);
        return KronConstructShifted(Mat_L, Mat_R, MatOut);
    }

    /** Same, for operator products whose total sector shift is not one of the Op_t values (e.g. Sp_i Sp_j on one side):
        the shifts are read from the matrices. */
    PetscErrorCode KronConstructShifted(const Mat& Mat_L, const Mat& Mat_R, Mat& MatOut)
    {
        if (!Mat_L || !Mat_R) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "KronConstruct: null operator.");
        extra_left = Mat_L; extra_right = Mat_R;
        std::vector<Hamiltonians::Term> one = {{1.0, OpSz, -1, OpSz, -1}};
        PetscErrorCode ierr = BuildPlan(nullptr, nullptr, one, MatOut);
        extra_left = nullptr; extra_right = nullptr;
        return ierr;
    }

private:
    static Mat OpOf(Block::SpinBase& blk, Op_t op, PetscInt site, const Mat& extra)
    {
        if (site < 0) return extra;
        return op == OpSz ? blk.Sz(site) : blk.Sp(site);      /* Sm(i) is read as Sp(i) transposed */
    }

    PetscErrorCode BuildPlan(const Mat& HL, const Mat& HR, const std::vector<Hamiltonians::Term>& TermsLR, Mat& MatOut, const bool distributed = false)
    {
        std::map<std::pair<int, PetscInt>, int32_t> li, ri;
        std::vector<dmrgx_secop> lops, rops;
        std::vector<std::vector<dmrgx_cell>> store;
        store.reserve(2 * TermsLR.size() + 4);
        auto add = [&](Block::SpinBase& blk, std::map<std::pair<int, PetscInt>, int32_t>& idx, std::vector<dmrgx_secop>& ops, Op_t op, PetscInt site, const Mat& extra) -> int32_t {
            auto key = std::make_pair((int)op, site);
            auto it = idx.find(key);
            if (it != idx.end()) return it->second;
            Mat m = OpOf(blk, op, site, extra);
            if (!m) return -1;
            store.emplace_back();
            dmrgx_secop so;
            if (m->transpose_of) m->transpose_of->to_secop(so, store.back(), true, -m->transpose_of->shift);
            else if (site >= 0 && op == OpSm) m->to_secop(so, store.back(), true, -1);   /* block operator Sp(site) read as Sm */
            else m->to_secop(so, store.back());
            ops.push_back(so);
            idx[key] = (int32_t)ops.size() - 1;
            return idx[key];
        };
        std::vector<dmrgx_term> terms;
        for (const Hamiltonians::Term& t : TermsLR) {
            const int32_t l = add(LeftBlock, li, lops, t.Iop, t.Isite, extra_left), r = add(RightBlock, ri, rops, t.Jop, t.Jsite, extra_right);
            if (l < 0 || r < 0) SETERRQ(mpi_comm, PETSC_ERR_ARG_CORRUPT, "Term refers to an operator that does not exist.");
            terms.push_back(dmrgx_term{t.a, l, r});
        }
        dmrgx_secop hl, hr;
        std::vector<dmrgx_cell> hls, hrs;
        if (HL) HL->to_secop(hl, hls);
        if (HR) HR->to_secop(hr, hrs);
        const std::vector<int32_t> ls = LeftBlock.Magnetization.Sizes32(), rs = RightBlock.Magnetization.Sizes32();
        std::vector<int32_t> bil, bir;
        for (const KronBlock_t& kb : KronBlocks) { bil.push_back((int32_t)std::get<1>(kb)); bir.push_back((int32_t)std::get<2>(kb)); }
        dmrgx_kron_desc d;
        d.left = dmrgx_sectors{(int32_t)ls.size(), ls.data()};
        d.right = dmrgx_sectors{(int32_t)rs.size(), rs.data()};
        d.nblocks = (int32_t)bil.size(); d.block_il = bil.data(); d.block_ir = bir.data();
        d.n_left_ops = (int32_t)lops.size(); d.n_right_ops = (int32_t)rops.size();
        d.left_ops = lops.data(); d.right_ops = rops.data();
        d.h_left = HL ? &hl : nullptr; d.h_right = HR ? &hr : nullptr;
        d.nterms = (int32_t)terms.size(); d.terms = terms.data();
        d.world_size = 1; d.rank = 0;
        if (distributed && dmrgx_host::WorldComm()) { d.world_size = dmrgx_host::WorldSize(); d.rank = dmrgx_host::WorldRank(); }
        dmrgx_kron_plan* plan = nullptr;
        if (dmrgx_kron_plan_create(&d, nullptr, &plan)) SETERRQ1(mpi_comm, 1, "dmrgx_kron_plan_create: %s", dmrgx_last_error());
        MatOut = std::make_shared<dmrgx_host::SectorMat>();
        MatOut->plan = plan;
        MatOut->shell_n = num_states;
        MatOut->plan_world = d.world_size;
        return 0;
    }

    MPI_Comm mpi_comm = PETSC_COMM_SELF;
    const PetscInt GlobIdx;
    std::vector<KronBlock_t> KronBlocks;
    std::vector<PetscReal> kb_list;
    std::vector<PetscInt> kb_size, kb_offset;
    std::map<std::tuple<PetscInt, PetscInt>, PetscInt> kb_map;
    PetscInt num_blocks = 0, num_states = 0;
    Block::SpinBase& LeftBlock;
    Block::SpinBase& RightBlock;
    FILE* fp_prealloc;
    PetscBool do_redistribute = PETSC_FALSE, do_shell = PETSC_TRUE;
    PetscReal ks_tol = 1.0e-16;
    Mat extra_left = nullptr, extra_right = nullptr;
};

/** y = H x with the device plan behind the shell matrix (the MATOP_MULT callback of the reference). */
inline PetscErrorCode MatMult_KronSumShell(Mat A, Vec x, Vec y)
{
    if (!A || !A->plan || !x || !y) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_CORRUPT, "MatMult_KronSumShell: not a shell matrix / null vector.");
    if (A->plan_world > 1) {
        /* striped plan: x (reference layout, replicated) -> rank-major stripes; every rank applies its stripe; one
           all-gather; back to the reference layout -- the VecScatter-to-all + local rows of the reference (src/DMRGKron.cpp:1833-1869) */
        dmrgx_kron_info I;
        if (dmrgx_kron_plan_info(A->plan, &I)) SETERRQ1(PETSC_COMM_SELF, 1, "dmrgx_kron_plan_info: %s", dmrgx_last_error());
        dmrgx_host::DevBuffer xs((size_t)I.vec_len, dmrgx_host::DevBuffer::device_only_t{}), ys((size_t)I.vec_len, dmrgx_host::DevBuffer::device_only_t{});
        if (dmrgx_memset_zero(xs.dev_uninitialised(), (size_t)I.vec_len * sizeof(double), nullptr) || dmrgx_memset_zero(ys.dev_uninitialised(), (size_t)I.vec_len * sizeof(double), nullptr) ||
            dmrgx_kron_vec_to_striped(A->plan, x->buf->dev_ro(), xs.dev_uninitialised(), nullptr) ||
            dmrgx_kron_apply(A->plan, xs.dev_ro(), ys.dev_uninitialised() + I.local_offset, nullptr) ||
            dmrgx_comm_allgather(dmrgx_host::WorldComm(), ys.dev_uninitialised(), I.seg_stride, nullptr) ||
            dmrgx_kron_vec_from_striped(A->plan, ys.dev_ro(), y->buf->dev(), nullptr))
            SETERRQ1(PETSC_COMM_SELF, 1, "striped MatMult: %s", dmrgx_last_error());
        return 0;
    }
    if (dmrgx_kron_apply(A->plan, x->buf->dev_ro(), y->buf->dev(), nullptr)) SETERRQ1(PETSC_COMM_SELF, 1, "dmrgx_kron_apply: %s", dmrgx_last_error());
    return 0;
}
inline PetscErrorCode MatMult(Mat A, Vec x, Vec y) { return MatMult_KronSumShell(A, x, y); }

/** Releases the plan of a shell matrix (call before MatDestroy, as with the reference). */
inline PetscErrorCode MatDestroy_KronSumShell(Mat* p_mat)
{
    if (p_mat && *p_mat && (*p_mat)->plan) { dmrgx_kron_plan_destroy((*p_mat)->plan); (*p_mat)->plan = nullptr; }
    return 0;
}

inline PetscErrorCode MatCreateVecs(const Mat& A, Vec* right, Vec* left)
{
    for (Vec* v : {right, left}) if (v) { *v = std::make_shared<dmrgx_host::VecImpl>(); (*v)->n = A->N(); (*v)->buf = std::make_shared<dmrgx_host::DevBuffer>((size_t)A->N()); }
    return 0;
}

/** Walks a range of superblock basis states, decoding the KronBlock, the sectors and the local/global indices on
    both sides. */
class KronBlocksIterator
{
public:
    KronBlocksIterator(const KronBlocks_t& KB, const PetscInt& GlobIdxStart, const PetscInt& GlobIdxEnd)
        : KB(KB), istart_(GlobIdxStart), iend_(GlobIdxEnd), idx_(GlobIdxStart)
    {
        if (istart_ != iend_) while (idx_ >= KB.kb_offset[blockidx_ + 1]) ++blockidx_;
    }
    PetscInt IdxStart() const { return istart_; }
    PetscInt IdxEnd() const { return iend_; }
    PetscInt Idx() const { return idx_; }
    PetscInt BlockIdx() const { return blockidx_; }
    PetscInt LocIdx() const { return idx_ - KB.kb_offset[blockidx_]; }
    PetscInt BlockStartIdx(const PetscInt& BlockShift) const { const PetscInt b = blockidx_ + BlockShift; return (b < 0 || b >= KB.num_blocks) ? -1 : KB.kb_offset[b]; }
    PetscInt BlockSize(const PetscInt& BlockShift) const { const PetscInt b = blockidx_ + BlockShift; return (b < 0 || b >= KB.num_blocks) ? -1 : KB.kb_size[b]; }
    bool Loop() const { return idx_ < iend_; }
    PetscInt Steps() const { return idx_ - istart_; }
    KronBlocksIterator& operator++()
    {
        ++idx_;
        updated_block = PETSC_FALSE;
        if (idx_ < KB.num_states && idx_ >= KB.kb_offset[blockidx_ + 1]) { ++blockidx_; updated_block = PETSC_TRUE; }
        return *this;
    }
    PetscInt BlockIdxLeft() const { return std::get<1>(KB.KronBlocks[blockidx_]); }
    PetscInt BlockIdxRight() const { return std::get<2>(KB.KronBlocks[blockidx_]); }
    PetscInt NumStatesRight() const { return KB.RightBlock.Magnetization.Sizes(BlockIdxRight()); }
    PetscInt LocIdxLeft() const { return LocIdx() / NumStatesRight(); }
    PetscInt LocIdxRight() const { return LocIdx() % NumStatesRight(); }
    PetscInt GlobalIdxLeft() const { return KB.LeftBlock.Magnetization.BlockIdxToGlobalIdx(BlockIdxLeft(), LocIdxLeft()); }
    PetscInt GlobalIdxRight() const { return KB.RightBlock.Magnetization.BlockIdxToGlobalIdx(BlockIdxRight(), LocIdxRight()); }
    PetscBool UpdatedBlock() const { return updated_block; }
private:
    const KronBlocks_t& KB;
    PetscInt istart_ = 0, iend_ = 0, idx_ = 0, blockidx_ = 0;
    PetscBool updated_block = PETSC_TRUE;
};

namespace dmrgx_host {

/** Host materialisation of a Kronecker-padded cell; only reached when a block with multi-state sectors is attached
    on the right (never in a sweep, where the added site has one state per sector).
    eye_on_right: out = cell (x) 1_n                                   -> (nr n) x (nc n), entry ((i n + r), (j n + r))
    else        : out = 1_n (x) [cell embedded at (r0,c0) of an R x C block] -> (n R) x (n C), entry (l R + r0 + i, l C + c0 + j) */
inline MatCell KronExpandHost(const MatCell& c, int32_t n, bool eye_on_right, int32_t R = 0, int32_t C = 0)
{
    MatCell o;
    o.nr = eye_on_right ? c.nr * n : n * R; o.nc = eye_on_right ? c.nc * n : n * C; o.ld = o.nc; o.kind = DMRGX_CELL_DENSE;
    o.buf = std::make_shared<DevBuffer>((size_t)o.nr * o.nc);
    double* d = o.buf->host();
    for (int32_t i = 0; i < c.nr; ++i)
        for (int32_t j = 0; j < c.nc; ++j) {
            const double v = (c.kind == DMRGX_CELL_DENSE) ? c.buf->host_ro()[c.off + (int64_t)i * c.ld + j] : (i == j ? c.scale : 0.0);
            for (int32_t r = 0; r < n; ++r) {
                if (eye_on_right) d[(int64_t)(i * n + r) * o.ld + (j * n + r)] = v;
                else d[(int64_t)(r * R + c.r0 + i) * o.ld + (r * C + c.c0 + j)] = v;
            }
        }
    return o;
}

}  // namespace dmrgx_host

/** BlockOut = LeftBlock (x) RightBlock: sectors merged by total Sz, every site operator padded with the identity of
    the other block, and H_out = H_L (x) 1 + 1 (x) H_R + the inter-block terms among the first nsites_out sites. */
inline PetscErrorCode KronEye_Explicit(Block::SpinBase& LeftBlock, Block::SpinBase& RightBlock,
                                       const std::vector<Hamiltonians::Term>& Terms, Block::SpinBase& BlockOut)
{
    using namespace dmrgx_host;
    PetscErrorCode ierr = 0;
    if (!LeftBlock.Initialized()) SETERRQ(PETSC_COMM_SELF, 1, "Left input block not initialized.");
    if (!RightBlock.Initialized()) SETERRQ(PETSC_COMM_SELF, 1, "Right input block not initialized.");
    MPI_Comm mpi_comm = LeftBlock.MPIComm();
    if (mpi_comm != RightBlock.MPIComm()) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "Input blocks must have the same communicator.");
    ierr = LeftBlock.CheckOperators(); CHKERRQ(ierr);  ierr = LeftBlock.CheckSectors(); CHKERRQ(ierr);  ierr = LeftBlock.CheckOperatorBlocks(); CHKERRQ(ierr);
    ierr = RightBlock.CheckOperators(); CHKERRQ(ierr); ierr = RightBlock.CheckSectors(); CHKERRQ(ierr); ierr = RightBlock.CheckOperatorBlocks(); CHKERRQ(ierr);

    KronBlocks_t KB(LeftBlock, RightBlock, {}, NULL, -1);
    const PetscInt nl = LeftBlock.NumSites(), nr = RightBlock.NumSites(), nout = nl + nr;
    const PetscInt nb = KB.size();
    /* merge equal-QN KronBlocks (already adjacent after the stable sort) into the output sectors */
    std::vector<PetscReal> QN_List; std::vector<PetscInt> QN_Size;
    std::vector<int32_t> sec((size_t)nb), sub((size_t)nb);
    for (PetscInt k = 0; k < nb; ++k) {
        if (QN_List.empty() || KB.QN(k) < QN_List.back()) { QN_List.push_back(KB.QN(k)); QN_Size.push_back(0); }
        sec[k] = (int32_t)QN_List.size() - 1; sub[k] = (int32_t)QN_Size.back();
        QN_Size.back() += KB.Sizes(k);
    }
    PetscInt tot = 0; for (PetscInt s : QN_Size) tot += s;
    if (tot != LeftBlock.NumStates() * RightBlock.NumStates()) SETERRQ2(mpi_comm, 1, "Mismatch in number of states. Expected %lld. Got %lld.", LLD(LeftBlock.NumStates() * RightBlock.NumStates()), LLD(tot));
    for (const Hamiltonians::Term& t : Terms)
        if (t.Isite >= nout || t.Jsite >= nout) SETERRQ3(mpi_comm, 1, "Term indices must be less than %lld. Got %lld and %lld.", LLD(nout), LLD(t.Isite), LLD(t.Jsite));

    ierr = BlockOut.Initialize(mpi_comm, nout, QN_List, QN_Size, PETSC_FALSE); CHKERRQ(ierr);
    const std::vector<int32_t> out_sizes = BlockOut.Magnetization.Sizes32();
    const QuantumNumbers& ML = LeftBlock.Magnetization; const QuantumNumbers& MR = RightBlock.Magnetization;

    /* ---- site operators: views / identity cells ------------------------------------------------------------- */
    for (int side = 0; side < 2; ++side) {
        Block::SpinBase& blk = side == 0 ? LeftBlock : RightBlock;
        for (PetscInt isite = 0; isite < blk.NumSites(); ++isite)
            for (Op_t op : BasicOpTypes) {
                const Mat src = (op == OpSz) ? blk.Sz(isite) : blk.Sp(isite);
                if (!src) { BlockOut.SetOpsPruned(); continue; }      /* pruned site (Block::PruneOperators): stays absent in the enlarged block */
                Mat out = std::make_shared<SectorMat>();
                out->shift = op; out->sizes = out_sizes;
                for (PetscInt k = 0; k < nb; ++k) {
                    const PetscInt IL = KB.LeftIdx(k), IR = KB.RightIdx(k);
                    const PetscInt kc = side == 0 ? KB.Map(IL + op, IR) : KB.Map(IL, IR + op);
                    if (kc < 0) continue;
                    const int32_t n = (int32_t)(side == 0 ? MR.Sizes(IR) : ML.Sizes(IL));   /* size of the identity factor */
                    for (const MatCell& c : src->cells) {
                        if (c.q != (side == 0 ? IL : IR)) continue;
                        MatCell o;
                        if (side == 0) {                                                         /* cell (x) 1_n */
                            if (n == 1) o = c;                                                   /* pure view of the same buffer */
                            else if (c.kind == DMRGX_CELL_IDENT) { o = c; o.nr = c.nr * n; o.nc = c.nc * n; }
                            else o = KronExpandHost(c, n, true);
                            o.r0 = sub[k] + c.r0 * n; o.c0 = sub[kc] + c.c0 * n;
                        } else {                                                                 /* 1_n (x) cell */
                            const int32_t Rr = (int32_t)MR.Sizes(IR), Rc = (int32_t)MR.Sizes(IR + op);
                            if (n == 1) { o = c; o.r0 = sub[k] + c.r0; o.c0 = sub[kc] + c.c0; }
                            else if (Rr == 1 && Rc == 1) {                                       /* 1_n (x) [v] = v 1_n */
                                o.kind = DMRGX_CELL_IDENT; o.nr = o.nc = n;
                                o.scale = (c.kind == DMRGX_CELL_DENSE) ? c.buf->host_ro()[c.off] : c.scale;
                                o.r0 = sub[k]; o.c0 = sub[kc];
                            }
                            else { o = KronExpandHost(c, n, false, Rr, Rc); o.r0 = sub[k]; o.c0 = sub[kc]; }
                        }
                        o.q = sec[k];
                        if (o.kind == DMRGX_CELL_IDENT && o.scale == 0.0) continue;
                        out->cells.push_back(o);
                    }
                }
                BlockOut.SetOp(op, isite + (side == 0 ? 0 : nl), out);
            }
    }

    /* ---- block Hamiltonian: dense sector blocks assembled on the device ----------------------------------------- */
    std::vector<Hamiltonians::Term> TermsLR;
    ierr = KB.ClassifyTerms(Terms, TermsLR); CHKERRQ(ierr);
    if (!LeftBlock.H && !RightBlock.H && TermsLR.empty()) { BlockOut.H = nullptr; return 0; }
    /* all sector blocks in one device allocation, zeroed by one launch (one allocation and one launch per block before) */
    Mat Hout = std::make_shared<SectorMat>();
    Hout->shift = 0; Hout->sizes = out_sizes;
    size_t htotal = 0;
    for (int32_t sz : out_sizes) htotal += (size_t)sz * (size_t)sz;
    std::shared_ptr<dmrgx_host::DevBuffer> harena;
    try { harena = std::make_shared<dmrgx_host::DevBuffer>(htotal, dmrgx_host::DevBuffer::device_only_t{}); } catch (const std::exception& e) { SETERRQ1(mpi_comm, PETSC_ERR_MEM, "KronEye_Explicit: %s", e.what()); }
    if (htotal && dmrgx_memset_zero(harena->dev_uninitialised(), htotal * sizeof(double), nullptr)) SETERRQ1(mpi_comm, 1, "%s", dmrgx_last_error());
    std::vector<dmrgx_axpy_task> tasks;
    std::vector<double*> blockptr(out_sizes.size());
    {
        size_t cursor = 0;
        for (int32_t q = 0; q < (int32_t)out_sizes.size(); ++q) {
            dmrgx_host::MatCell c;
            c.q = q; c.nr = out_sizes[(size_t)q]; c.nc = c.nr; c.ld = c.nc; c.buf = harena; c.off = (int64_t)cursor;
            blockptr[(size_t)q] = harena->dev_uninitialised() + cursor;
            cursor += (size_t)c.nr * (size_t)c.nc;
            Hout->cells.push_back(c);
        }
    }
    auto push = [&](PetscInt k, PetscInt kc, const MatCell& c, bool tr, double alpha, int32_t rmul) {
        /* out[sub[k] + r0.., sub[kc] + c0..] += alpha * cell (transposed view when tr) ; rmul: 1 (only 1-state right sectors) */
        const int32_t q = sec[k];
        const int64_t ld = out_sizes[q];
        const int32_t r0 = tr ? c.c0 : c.r0, c0 = tr ? c.r0 : c.c0, nrr = tr ? c.nc : c.nr, ncc = tr ? c.nr : c.nc;
        dmrgx_axpy_task t;
        t.dst = blockptr[q] + (int64_t)(sub[k] + r0 * rmul) * ld + (sub[kc] + c0 * rmul);
        t.dst_base = blockptr[q] + (int64_t)sub[k] * ld + sub[kc];      /* tasks into one (k,kc) sub-block are ordered */
        if (c.kind == DMRGX_CELL_DENSE) { t.src = c.buf->dev_ro() + c.off; t.alpha = alpha; }
        else { t.src = nullptr; t.alpha = alpha * c.scale; }                 /* scaled identity: diagonal add */
        t.ldd = ld; t.lds = c.ld; t.nr = nrr; t.nc = ncc; t.transposed = tr ? 1 : 0;
        tasks.push_back(t);
    };
    for (PetscInt k = 0; k < nb; ++k) {
        const PetscInt IL = KB.LeftIdx(k), IR = KB.RightIdx(k);
        const bool one_r = MR.Sizes(IR) == 1, one_l = ML.Sizes(IL) == 1;
        if (LeftBlock.H) for (const MatCell& c : LeftBlock.H->cells) if (c.q == IL) {
            if (!one_r) SETERRQ(mpi_comm, PETSC_ERR_SUP, "KronEye_Explicit: H_L (x) 1 needs one-state right sectors.");
            push(k, k, c, false, 1.0, 1);
        }
        if (RightBlock.H) for (const MatCell& c : RightBlock.H->cells) if (c.q == IR) {
            if (c.kind != DMRGX_CELL_DENSE) SETERRQ(mpi_comm, PETSC_ERR_SUP, "KronEye_Explicit: unsupported H_R cell.");
            bool zero = true;                                          /* the added site's own Hamiltonian is zero */
            const double* h = c.buf->host_ro();
            for (int32_t i = 0; i < c.nr && zero; ++i) for (int32_t j = 0; j < c.nc; ++j) if (h[c.off + (int64_t)i * c.ld + j] != 0.0) { zero = false; break; }
            if (zero) continue;
            if (!one_l) SETERRQ(mpi_comm, PETSC_ERR_SUP, "KronEye_Explicit: a non-zero 1 (x) H_R needs one-state left sectors.");
            push(k, k, c, false, 1.0, 1);
        }
        for (const Hamiltonians::Term& t : TermsLR) {
            const PetscInt sA = (t.Iop == OpSz) ? 0 : (PetscInt)t.Iop, sB = (t.Jop == OpSz) ? 0 : (PetscInt)t.Jop;
            const PetscInt kc = KB.Map(IL + sA, IR + sB);
            if (kc < 0 || sec[kc] != sec[k]) continue;
            if (!one_r || MR.Sizes(IR + sB) != 1) SETERRQ(mpi_comm, PETSC_ERR_SUP, "KronEye_Explicit: inter-block terms need one-state right sectors (added site).");
            /* right factor: the 1x1 block (IR -> IR+sB) of the site operator */
            const Mat B = (t.Jop == OpSz) ? RightBlock.Sz(t.Jsite) : RightBlock.Sp(t.Jsite);
            const Mat A = (t.Iop == OpSz) ? LeftBlock.Sz(t.Isite) : LeftBlock.Sp(t.Isite);
            if (!A || !B) SETERRQ2(mpi_comm, PETSC_ERR_ARG_CORRUPT, "KronEye_Explicit: the term between sites %lld and %lld needs an operator that is not resident (pruned).", LLD(t.Isite), LLD(t.Jsite));
            const bool trB = (t.Jop == OpSm);
            double b = 0.0;
            for (const MatCell& c : B->cells) {
                const int32_t rowsec = trB ? c.q + B->shift : c.q;     /* transposed view: block (q+1 -> q) */
                if (rowsec != IR) continue;
                if (c.kind == DMRGX_CELL_DENSE) b += c.buf->host_ro()[c.off]; else b += c.scale;
            }
            if (b == 0.0) continue;
            const bool trA = (t.Iop == OpSm);
            for (const MatCell& c : A->cells) {
                const int32_t rowsec = trA ? c.q + A->shift : c.q;
                if (rowsec != IL) continue;
                push(k, kc, c, trA, t.a * b, 1);
            }
        }
    }
    if (!tasks.empty() && dmrgx_cells_axpy((int32_t)tasks.size(), tasks.data(), nullptr)) SETERRQ1(mpi_comm, 1, "dmrgx_cells_axpy: %s", dmrgx_last_error());
    BlockOut.H = Hout;
    return 0;
}

#endif
