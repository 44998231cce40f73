/** @file SectorMat.hpp
    The engine's `Mat` and `Vec`: handles to sector-blocked operator cells / superblock vectors whose storage lives in
    HBM behind the C ABI (include/dmrgx.h).  They take the place of PETSc's Mat/Vec in the DMRG.x operator API
    (reference include/DMRGBlock.hpp:79-434).  An operator is a list of dense or scaled-identity cells inside its
    (sector q -> sector q+shift) blocks -- exactly the dmrgx_cell description the kernels consume, so building a
    superblock plan or a rotation needs no conversion.  A host mirror exists only on demand (fixtures, single-site
    operators, tests); all arithmetic happens on the device. */
#ifndef DMRGX_SECTORMAT_HPP
#define DMRGX_SECTORMAT_HPP

#include <memory>
#include <vector>
#include <algorithm>
#include "petsc_compat.hpp"
#include "dmrgx.h"

namespace dmrgx_host {

/** f64 buffer with a lazily synchronised host mirror. */
class DevBuffer {
public:
    /** n zeros.  The host mirror is only materialised when somebody asks for a host pointer: buffers that are filled on the device
        (eigenvectors, enlarged Hamiltonians: 30-70 MB at m = 2048) used to cost a calloc + page faults of that size per step. */
    explicit DevBuffer(size_t n) : n_(n), where_(HOST) {}
    /** device-resident buffer whose content is produced by a kernel: no host mirror is allocated until someone asks for it */
    struct device_only_t {};
    DevBuffer(size_t n, device_only_t) : n_(n), where_(DEVICE) { alloc_dev(); }
    /** n elements of `owner`'s device storage starting at `offset`: a device-only view that keeps the owner alive (one arena,
        one allocation, many sector blocks) */
    DevBuffer(const std::shared_ptr<DevBuffer>& owner, size_t offset, size_t n) : n_(n), d_(owner->dev_uninitialised() + offset), where_(DEVICE), owner_(owner) {}
    DevBuffer(const DevBuffer&) = delete;
    DevBuffer& operator=(const DevBuffer&) = delete;
    ~DevBuffer() { if (d_ && !owner_) dmrgx_free(d_); }
    size_t size() const { return n_; }
    /** host pointer for reading and writing (device copy becomes stale) */
    double* host() { sync_host(); where_ = HOST; return h_.data(); }
    const double* host_ro() { sync_host(); return h_.data(); }
    /** device pointer for reading and writing (host copy becomes stale); throws without a GPU */
    double* dev() { sync_dev(); where_ = DEVICE; return d_; }
    const double* dev_ro() { sync_dev(); return d_; }
    /** device pointer without uploading the (all-zero / don't-care) host content */
    double* dev_uninitialised() {
        if (!d_) alloc_dev();
        where_ = DEVICE;
        return d_;
    }
private:
    enum Where { HOST, DEVICE, BOTH };
    void alloc_dev() {
        void* p = nullptr;
        if (dmrgx_malloc(&p, std::max<size_t>(n_, 1) * sizeof(double))) throw std::runtime_error(std::string("dmrgx_malloc: ") + dmrgx_last_error());
        d_ = static_cast<double*>(p);
    }
    void sync_dev() {
        if (!d_) alloc_dev();
        if (where_ == HOST) {
            if (h_.size() != n_) {                /* still the constructor's zeros: no host copy exists, none is made */
                if (n_ && dmrgx_memset_zero(d_, n_ * sizeof(double), nullptr)) throw std::runtime_error(dmrgx_last_error());
            } else {
                if (n_ && dmrgx_memcpy_h2d(d_, h_.data(), n_ * sizeof(double), nullptr)) throw std::runtime_error(dmrgx_last_error());
                if (dmrgx_stream_sync(nullptr)) throw std::runtime_error(dmrgx_last_error());
            }
            where_ = BOTH;
        }
    }
    void sync_host() {
        if (h_.size() != n_) h_.assign(n_, 0.0);          /* (HOST / BOTH without a mirror: the content is the constructor's zeros) */
        if (where_ == DEVICE) {
            if (n_ && dmrgx_memcpy_d2h(h_.data(), d_, n_ * sizeof(double), nullptr)) throw std::runtime_error(dmrgx_last_error());
            where_ = BOTH;
        }
    }
    size_t n_;
    std::vector<double> h_;
    double* d_ = nullptr;
    Where where_;
    std::shared_ptr<DevBuffer> owner_;
};

struct MatCell {
    int32_t q = 0, r0 = 0, c0 = 0, nr = 0, nc = 0;
    int32_t kind = DMRGX_CELL_DENSE;
    double scale = 0.0;
    std::shared_ptr<DevBuffer> buf;   /**< DENSE: storage (possibly shared with the block this one was enlarged from) */
    int64_t off = 0, ld = 0;          /**< element (r0,c0) is buf[off], row stride ld */
};

/** Truncation as a sector-block rotation (the engine's RotMatT): new sector a keeps kept[a] states of old sector
    old_sector[a]; rt[a] holds those kept[a] eigenvectors as rows (kept[a] x n_old, row-major) -- the non-zero
    blocks of the reference's RotMatT (include/DMRGBlockContainer.hpp:2032-2054). */
struct BasisRotation {
    std::vector<int32_t> old_sizes;
    std::vector<int32_t> old_sector, kept;
    std::vector<std::shared_ptr<DevBuffer>> rt;
};

/** Sector-blocked square operator (or, when `plan` is set, a matrix-free superblock Hamiltonian; or, when `rot` is
    set, a rotation matrix RotMatT). */
class SectorMat {
public:
    std::shared_ptr<SectorMat> transpose_of;   /**< set for Sm(i): a transposed view of Sp(i), never materialised */
    std::shared_ptr<BasisRotation> rot;        /**< set for RotMatT */
    int32_t shift = 0;
    std::vector<int32_t> sizes;       /**< sector sizes of the block basis */
    std::vector<MatCell> cells;
    dmrgx_kron_plan* plan = nullptr;  /**< shell matrix: the HIP plan that applies it */
    PetscInt shell_n = 0;
    int32_t plan_world = 1;           /**< > 1: the plan is striped over the ranks of the communicator */

    PetscInt N() const { if (plan) return shell_n; PetscInt n = 0; for (int32_t s : sizes) n += s; return n; }
    std::vector<PetscInt> offsets() const { std::vector<PetscInt> o(sizes.size() + 1, 0); for (size_t i = 0; i < sizes.size(); ++i) o[i + 1] = o[i] + sizes[i]; return o; }

    /** One zero-initialised dense cell per existing sector block: the form block initialisers hand out
        (InitSingleSiteOperator + preallocation in the reference, src/DMRGBlock.cpp:140-147). */
    static std::shared_ptr<SectorMat> Dense(int32_t shift, const std::vector<int32_t>& sizes) {
        auto m = std::make_shared<SectorMat>();
        m->shift = shift; m->sizes = sizes;
        const int32_t ns = (int32_t)sizes.size();
        for (int32_t q = 0; q < ns; ++q) {
            const int32_t qc = q + shift;
            if (qc < 0 || qc >= ns) continue;
            MatCell c;
            c.q = q; c.nr = sizes[q]; c.nc = sizes[qc]; c.ld = c.nc;
            c.buf = std::make_shared<DevBuffer>((size_t)c.nr * c.nc);
            m->cells.push_back(c);
        }
        return m;
    }

    /** Locate global (row, col): sector indices and in-block coordinates; false if outside the shift's blocks. */
    bool locate(PetscInt row, PetscInt col, int32_t& q, int32_t& i, int32_t& j) const {
        const auto o = offsets();
        const int32_t ns = (int32_t)sizes.size();
        q = -1;
        for (int32_t s = 0; s < ns; ++s) if (row >= o[s] && row < o[s + 1]) q = s;
        if (q < 0) return false;
        const int32_t qc = q + shift;
        if (qc < 0 || qc >= ns || col < o[qc] || col >= o[qc + 1]) return false;
        i = (int32_t)(row - o[q]); j = (int32_t)(col - o[qc]);
        return true;
    }
    /** Host-side element access (fixtures / tests).  set() needs a dense cell covering the entry. */
    double get(PetscInt row, PetscInt col) const {
        int32_t q, i, j;
        if (!locate(row, col, q, i, j)) return 0.0;
        double v = 0.0;
        for (const MatCell& c : cells) {
            if (c.q != q || i < c.r0 || i >= c.r0 + c.nr || j < c.c0 || j >= c.c0 + c.nc) continue;
            if (c.kind == DMRGX_CELL_DENSE) v += c.buf->host_ro()[c.off + (int64_t)(i - c.r0) * c.ld + (j - c.c0)];
            else if (i - c.r0 == j - c.c0) v += c.scale;
        }
        return v;
    }
    PetscErrorCode set(PetscInt row, PetscInt col, double v) {
        int32_t q, i, j;
        if (!locate(row, col, q, i, j)) return PETSC_ERR_ARG_OUTOFRANGE;      /* entry outside the operator's sector blocks */
        for (MatCell& c : cells) {
            if (c.q != q || c.kind != DMRGX_CELL_DENSE || i < c.r0 || i >= c.r0 + c.nr || j < c.c0 || j >= c.c0 + c.nc) continue;
            c.buf->host()[c.off + (int64_t)(i - c.r0) * c.ld + (j - c.c0)] = v;
            return 0;
        }
        return PETSC_ERR_ARG_OUTOFRANGE;
    }
    std::vector<double> dense_row(PetscInt row) const {
        const PetscInt n = N();
        std::vector<double> r((size_t)n, 0.0);
        const auto o = offsets();
        int32_t q = -1;
        for (size_t s = 0; s < sizes.size(); ++s) if (row >= o[s] && row < o[s + 1]) q = (int32_t)s;
        const int32_t qc = q + shift;
        if (q < 0 || qc < 0 || qc >= (int32_t)sizes.size()) return r;
        for (PetscInt col = o[qc]; col < o[qc + 1]; ++col) r[(size_t)col] = get(row, col);
        return r;
    }
    /** dmrgx_secop view of this operator (device pointers; uploads host-resident cells). */
    void to_secop(dmrgx_secop& op, std::vector<dmrgx_cell>& storage, bool transposed = false, int32_t shift_as_used = 0) {
        storage.clear();
        for (MatCell& c : cells) {
            dmrgx_cell d;
            d.row_sector = c.q; d.r0 = c.r0; d.c0 = c.c0; d.nr = c.nr; d.nc = c.nc; d.kind = c.kind; d.scale = c.scale;
            d.data = (c.kind == DMRGX_CELL_DENSE) ? c.buf->dev_ro() + c.off : nullptr;
            d.ld = c.ld;
            storage.push_back(d);
        }
        op.shift = transposed ? shift_as_used : shift;
        op.transposed = transposed ? 1 : 0;
        op.ncells = (int32_t)storage.size();
        op.cells = storage.data();
    }
    void release() { cells.clear(); }
};

typedef std::shared_ptr<SectorMat> Mat;

/** Superblock vector (device). */
struct VecImpl { std::shared_ptr<DevBuffer> buf; PetscInt n = 0; };
typedef std::shared_ptr<VecImpl> Vec;

inline PetscErrorCode MatDestroy(Mat* m) { if (m && *m) { (*m)->release(); m->reset(); } return 0; }
inline PetscErrorCode VecDestroy(Vec* v) { if (v) v->reset(); return 0; }
inline PetscErrorCode MatGetSize(const Mat& m, PetscInt* M, PetscInt* N) { if (!m) return PETSC_ERR_ARG_CORRUPT; *M = *N = m->N(); return 0; }

/** Dense form of an operator: one device-resident dense cell per sector block (q -> q+shift), the cells of `src`
    accumulated into it on the device (dmrgx_cells_axpy: dense, identity and -- for a transposed view such as Sm(i) --
    transposed sources).  Used for operator products inside one block basis (correlators,
    include/DMRGBlockContainer.hpp:2333-2410 of the reference); never on the superblock path. */
inline PetscErrorCode SectorMatDensify(const Mat& src_in, Mat& out)
{
    if (!src_in) return PETSC_ERR_ARG_CORRUPT;
    const bool tr = (bool)src_in->transpose_of;
    const Mat src = tr ? src_in->transpose_of : src_in;
    const int32_t shift = tr ? -src->shift : src->shift;
    const int32_t ns = (int32_t)src->sizes.size();
    out = std::make_shared<SectorMat>();
    out->shift = shift; out->sizes = src->sizes;
    std::vector<int32_t> cell_of_sector((size_t)ns, -1);
    size_t total = 0;
    for (int32_t q = 0; q < ns; ++q) {
        const int32_t qc = q + shift;
        if (qc < 0 || qc >= ns) continue;
        total += (size_t)src->sizes[q] * (size_t)src->sizes[qc];
    }
    std::shared_ptr<DevBuffer> arena;
    try { arena = std::make_shared<DevBuffer>(total, DevBuffer::device_only_t{}); } catch (const std::exception&) { return PETSC_ERR_MEM; }
    double* base = arena->dev_uninitialised();
    if (total && dmrgx_memset_zero(base, total * sizeof(double), nullptr)) return 1;
    size_t cursor = 0;
    for (int32_t q = 0; q < ns; ++q) {
        const int32_t qc = q + shift;
        if (qc < 0 || qc >= ns) continue;
        MatCell c;
        c.q = q; c.nr = src->sizes[q]; c.nc = src->sizes[qc]; c.ld = c.nc; c.buf = arena; c.off = (int64_t)cursor;
        cursor += (size_t)c.nr * c.nc;
        cell_of_sector[(size_t)q] = (int32_t)out->cells.size();
        out->cells.push_back(c);
    }
    std::vector<dmrgx_axpy_task> tasks;
    for (MatCell& c : src->cells) {
        const int32_t qdst = tr ? c.q + src->shift : c.q;           /* transposed: block (q -> q+s) lands in row sector q+s */
        if (qdst < 0 || qdst >= ns || cell_of_sector[(size_t)qdst] < 0) continue;
        const MatCell& d = out->cells[(size_t)cell_of_sector[(size_t)qdst]];
        double* blk = base + d.off;
        dmrgx_axpy_task t;
        t.dst_base = blk;
        t.ldd = d.ld;
        t.transposed = tr ? 1 : 0;
        if (tr) { t.dst = blk + (int64_t)c.c0 * d.ld + c.r0; t.nr = c.nc; t.nc = c.nr; }
        else    { t.dst = blk + (int64_t)c.r0 * d.ld + c.c0; t.nr = c.nr; t.nc = c.nc; }
        if (c.kind == DMRGX_CELL_DENSE) { t.src = c.buf->dev_ro() + c.off; t.lds = c.ld; t.alpha = 1.0; }
        else { t.src = nullptr; t.lds = 0; t.alpha = c.scale; }
        tasks.push_back(t);
    }
    if (!tasks.empty() && dmrgx_cells_axpy((int32_t)tasks.size(), tasks.data(), nullptr)) return 1;
    return 0;
}

/** C = A * B for two operators in dense form on the same block basis (shifts add); per row sector one MFMA GEMM. */
inline PetscErrorCode SectorMatMatMult(const Mat& A, const Mat& B, Mat& C)
{
    if (!A || !B || A->sizes != B->sizes) return PETSC_ERR_ARG_CORRUPT;
    const int32_t ns = (int32_t)A->sizes.size(), sA = A->shift, sB = B->shift;
    C = std::make_shared<SectorMat>();
    C->shift = sA + sB; C->sizes = A->sizes;
    size_t total = 0;
    for (int32_t q = 0; q < ns; ++q) { const int32_t qc = q + sA + sB; if (qc >= 0 && qc < ns) total += (size_t)A->sizes[q] * (size_t)A->sizes[qc]; }
    std::shared_ptr<DevBuffer> arena;
    try { arena = std::make_shared<DevBuffer>(total, DevBuffer::device_only_t{}); } catch (const std::exception&) { return PETSC_ERR_MEM; }
    double* base = arena->dev_uninitialised();
    if (total && dmrgx_memset_zero(base, total * sizeof(double), nullptr)) return 1;
    auto cell_at = [](const Mat& M, int32_t q) -> const MatCell* { for (const MatCell& c : M->cells) if (c.q == q) return &c; return nullptr; };
    size_t cursor = 0;
    std::vector<dmrgx_gemm_task> tasks;
    for (int32_t q = 0; q < ns; ++q) {
        const int32_t qa = q + sA, qc = qa + sB;
        if (qc < 0 || qc >= ns) continue;
        MatCell c;
        c.q = q; c.nr = A->sizes[q]; c.nc = A->sizes[qc]; c.ld = c.nc; c.buf = arena; c.off = (int64_t)cursor;
        cursor += (size_t)c.nr * c.nc;
        C->cells.push_back(c);
        if (qa < 0 || qa >= ns) continue;                            /* the intermediate sector does not exist: zero block */
        const MatCell* a = cell_at(A, q);
        const MatCell* b = cell_at(B, qa);
        if (!a || !b || c.nr == 0 || c.nc == 0 || a->nc == 0) continue;
        tasks.push_back(dmrgx_gemm_task{c.nr, c.nc, a->nc, 0, a->buf->dev_ro() + a->off, a->ld, b->buf->dev_ro() + b->off, b->ld, base + c.off, c.ld});
    }
    if (!tasks.empty() && dmrgx_dgemm_batch((int32_t)tasks.size(), tasks.data(), nullptr)) return 1;   /* one grouped launch */
    return 0;
}

}  // namespace dmrgx_host

using dmrgx_host::Mat;
using dmrgx_host::Vec;
using dmrgx_host::MatDestroy;
using dmrgx_host::VecDestroy;
using dmrgx_host::MatGetSize;

#endif
