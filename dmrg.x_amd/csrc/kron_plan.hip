// K1: matrix-free superblock Hamiltonian apply  y = H_sb x  on the target-Sz sector.
//
// Replaces KronSumConstructShell + MatMult_KronSumShell (reference src/DMRGKron.cpp:1706-1917).  The reference
// evaluates, for every row, sum_t a_t sum_l sum_r A_t[l,l'] B_t[r,r'] x[..] (unfactored, :1844-1864).  Here the
// same operator is applied in factored, operator-merged form per KronBlock k=(IL,IR):
//
//     Y_k = H_L[IL] X_k  +  X_k H_R[IR]^T  +  sum_g  Abar_g[IL->IL'] ( X_k' Bhat_g[IR->IR']^T )
//
// where g runs over groups of terms sharing one operator on one side (the side with fewer distinct operators),
// the other side's operators being pre-summed with their coefficients (Abar_g = sum_t a_t A_t) at plan time.
// Stage 1 computes T_{g,k} = X_k' Bhat^T, stage 2 accumulates all groups into Y_k; both stages are ONE launch
// each of the grouped MFMA-f64 GEMM (ggemm.hip) over host-built task tables, with operator zero-cells skipped
// and identity cells (new-site operators) turned into scaled copies.  X_k is the row-major n_L x n_R matrix at
// the KronBlock offset (reference include/DMRGKron.hpp:198-209, 603-612).
//
// Multi-GPU: the right index of every KronBlock is split into world_size contiguous stripes; rank r owns
// column stripe r of every Y_k.  A full vector is stored rank-major (segment r = all stripes of rank r), so one
// RCCL all-gather of equal-sized segments rebuilds x for the next apply.
#include "ggemm.h"
#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <tuple>
#include <cstring>
#include <memory>
#include <new>

namespace dmrgx {

namespace {

struct NCell {              // normalised operator cell (transpose folded in)
    int32_t q, r0, c0, nr, nc, kind;
    double scale;
    const double* data;     // caller's device memory (only read during plan creation)
    int64_t ld;
    bool tr;                // element (i,j) of the cell is data[j*ld + i]
};

struct PCell {              // plan-owned cell: dense data lives in the arena at `off` (row-major, ld = nc)
    int32_t q, r0, c0, nr, nc, kind;
    double scale;
    int64_t off;
};

struct CopyTask {           // dst[i*ldd + j] = (round 0) / += (later rounds) a * src(i,j): every contribution covers its whole destination cell
    int64_t dst_off;        // arena element offset
    const double* src;
    int64_t lds;
    int32_t nr, nc, ldd, tr;
    double a;
    int32_t round;          // contributions to the same destination are applied in separate launches
};

struct CopyTile { int32_t task, ti, tj, pad; };

// The column segments of the intermediates T_{g,k} that no right-operator cell reaches are read by stage 2 as zeros: they are the only part of
// the arena that is zeroed (round 5; rounds 1-4 zeroed the whole arena -- 1.4 GB at m = 2048 -- and added every operator into it)
struct ZeroRect { int64_t off; int32_t ld, nr, nc, pad; };
__global__ void __launch_bounds__(256) zero_rects_kernel(const ZeroRect* __restrict__ rects, double* __restrict__ arena)
{
    const ZeroRect r = rects[blockIdx.y];
    const int64_t n = (int64_t)r.nr * r.nc;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) arena[r.off + (e / r.nc) * r.ld + e % r.nc] = 0.0;
}

__global__ void __launch_bounds__(256)
cell_copy_kernel(const CopyTile* __restrict__ tiles, const CopyTask* __restrict__ tasks, double* __restrict__ arena)
{
    __shared__ double buf[32][33];
    const CopyTile t = tiles[blockIdx.x];
    const CopyTask k = tasks[t.task];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int i0 = t.ti * 32, j0 = t.tj * 32;
    double* dst = arena + k.dst_off;
    if (!k.tr) {
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + tx;
            if (i < k.nr && j < k.nc) { const double v = k.a * k.src[(size_t)i * k.lds + j]; double& o = dst[(size_t)i * k.ldd + j]; o = k.round == 0 ? v : o + v; }
        }
    } else {
        // dst(i,j) = src[j*lds + i]: read coalesced along i, transpose through LDS
        for (int r = ty; r < 32; r += 8) {
            const int j = j0 + r, i = i0 + tx;
            buf[r][tx] = (i < k.nr && j < k.nc) ? k.src[(size_t)j * k.lds + i] : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + tx;
            if (i < k.nr && j < k.nc) { const double v = k.a * buf[tx][r]; double& o = dst[(size_t)i * k.ldd + j]; o = k.round == 0 ? v : o + v; }
        }
    }
}

// striped <-> reference vector layout
struct LayoutSeg { int64_t ref_off, full_off; int32_t nrow, ncol, ref_ld, full_ld; };
__global__ void layout_copy_kernel(const LayoutSeg* __restrict__ segs, const double* __restrict__ src, double* __restrict__ dst, int to_striped)
{
    const LayoutSeg s = segs[blockIdx.y];
    const int64_t n = (int64_t)s.nrow * s.ncol;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / s.ncol, j = e % s.ncol;
        const int64_t a = s.ref_off + i * s.ref_ld + j, b = s.full_off + i * s.full_ld + j;
        if (to_striped) dst[b] = src[a]; else dst[a] = src[b];
    }
}

// Split-K fix-up: for every split output block, y_block += slab_0 + slab_1 + ... in fixed order (bit-reproducible,
// no atomics).  Slabs are compact M x N copies (ld = N) stored back to back in the arena.
struct RedTask { int64_t dst_off, slab_off; int32_t ldc, M, N, nslab; };
struct RedTile { int32_t task, chunk; };
constexpr int RED_CHUNK = 2048;
__global__ __launch_bounds__(256) void slab_reduce_kernel(const RedTile* __restrict__ tiles, const RedTask* __restrict__ tasks,
                                                          double* __restrict__ y, const double* __restrict__ arena)
{
    const RedTile t = tiles[blockIdx.x];
    const RedTask k = tasks[t.task];
    const uint32_t mn = (uint32_t)k.M * (uint32_t)k.N, N = (uint32_t)k.N;     // a split block has < 2^31 elements (host check)
    const double* sl = arena + k.slab_off;
    double* yb = y + k.dst_off;
    constexpr int NE = RED_CHUNK / 256;
    const uint32_t e0 = (uint32_t)t.chunk * RED_CHUNK + threadIdx.x;
    double v[NE];
    double* d[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const uint32_t e = min(e0 + (uint32_t)i * 256u, mn - 1);        // clamped: every lane loads, only valid ones store
        const uint32_t row = e / N, col = e - row * N;
        d[i] = yb + (size_t)row * k.ldc + col;
        v[i] = *d[i];
    }
    for (int s = 0; s < k.nslab; ++s) {                                  // slab-major: NE independent loads in flight per step,
        const double* p = sl + (size_t)s * mn;                           // each element still summed in slab order
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] += p[min(e0 + (uint32_t)i * 256u, mn - 1)];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) if (e0 + (uint32_t)i * 256u < mn) *d[i] = v[i];
}

// ---- diagonal of the superblock Hamiltonian (preconditioner of the generalized-Davidson solver) --------------------------
// diag(H)[(l, r) of KronBlock k] = sum_t dA_t[l] * dB_t[r] over the "diagonal terms": H_L (x) 1, 1 (x) H_R and every merged
// operator pair with sector shift 0 (the Sz Sz terms).  dA_t / dB_t are the diagonals of the plan's own operator copies.
struct DiagSrc { int64_t off; int32_t ld, n, round; int64_t dst; double scale; };     // off < 0: scaled identity
struct DiagSeg { int64_t out_off; int32_t nrow, ncol; int64_t l0, r0; };
__global__ void __launch_bounds__(256) diag_gather_kernel(const DiagSrc* __restrict__ src, int nsrc, int round, const double* __restrict__ arena, double* __restrict__ dvec)
{
    const int t = blockIdx.x;
    if (t >= nsrc) return;
    const DiagSrc s = src[t];
    if (s.round != round) return;
    for (int i = threadIdx.x; i < s.n; i += 256) dvec[s.dst + i] += s.off >= 0 ? arena[s.off + (int64_t)i * (s.ld + 1)] : s.scale;
}
__global__ void __launch_bounds__(256) diag_fill_kernel(const DiagSeg* __restrict__ segs, const double* __restrict__ dA, const double* __restrict__ dB,
                                                         int nterms, int64_t NL, int64_t NR, double* __restrict__ out)
{
    const DiagSeg s = segs[blockIdx.y];
    const int64_t n = (int64_t)s.nrow * s.ncol;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t l = e / s.ncol, r = e % s.ncol;
        double d = 0.0;
        for (int t = 0; t < nterms; ++t) d += dA[(int64_t)t * NL + s.l0 + l] * dB[(int64_t)t * NR + s.r0 + r];
        out[s.out_off + e] = d;
    }
}

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

// Bases selectable by a task-table entry: the plan's arena, the apply's x and y.
enum : int32_t { BASE_ARENA = 0, BASE_X = 1, BASE_Y = 2 };

namespace dmrgx {
// Task tables are built with element offsets + base selectors and patched into absolute pointers per apply by
// a tiny kernel, so the ggemm kernel itself only ever sees plain pointers.
struct RelProd { int64_t a_off, b_off; int32_t lda, ldb, K, kind; double alpha; int32_t a_base, b_base; };
struct RelGroup { int64_t c_off; int32_t c_base, ldc, M, N, prod_begin, prod_end, n_axpy, accumulate; };

__global__ void patch_tables_kernel(const RelProd* __restrict__ rp, GProd* __restrict__ gp, int np,
                                    const RelGroup* __restrict__ rg, GGroup* __restrict__ gg, int ng,
                                    double* arena, const double* x, double* y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double* bases[3] = {arena, x, y};
    if (i < np) {
        const RelProd r = rp[i];
        GProd g;
        g.A = bases[r.a_base] + r.a_off; g.B = bases[r.b_base] + r.b_off;
        g.lda = r.lda; g.ldb = r.ldb; g.K = r.K; g.kind = r.kind; g.alpha = r.alpha;
        gp[i] = g;
    }
    if (i < ng) {
        const RelGroup r = rg[i];
        GGroup g;
        g.C = const_cast<double*>(bases[r.c_base]) + r.c_off;
        g.ldc = r.ldc; g.M = r.M; g.N = r.N; g.prod_begin = r.prod_begin; g.prod_end = r.prod_end;
        g.n_axpy = r.n_axpy; g.accumulate = r.accumulate;
        gg[i] = g;
    }
}
}  // namespace dmrgx

struct dmrgx_kron_plan {
    int32_t world = 1, rank = 0;
    dmrgx_kron_info info{};
    DevBuf arena;                       // operators + intermediates
    DevBuf d_tables;                    // one upload holds every table below (the DevBufs are views into it); declared first: destroyed last
    DevBuf d_rprods, d_rgroups;         // relative tables (stage 1 then stage 2, one array)
    // Patched (absolute-pointer) task tables, one set per (x, y) pair seen: a Lanczos solve applies the plan to the same
    // ncv+1 basis vectors cycle after cycle, so after the first cycle no apply has to re-patch (one launch less per step).
    struct Patched { const double* x = nullptr; double* y = nullptr; DevBuf prods, groups; };
    std::vector<std::unique_ptr<Patched>> patched;
    size_t patched_next = 0;            // round-robin victim once PATCHED_MAX sets exist
    static constexpr size_t PATCHED_MAX = 24;
    DevBuf d_tiles1, d_tiles2, d_tiles1b, d_tiles2b;
    int32_t nprods = 0, ngroups = 0, ntiles1 = 0, ntiles2 = 0, ntiles1b = 0, ntiles2b = 0;
    DevBuf d_layout;
    int32_t nlayout = 0;
    DevBuf d_red_tasks, d_red_tiles;    // split-K fix-up tables
    int32_t n_red_tiles = 0;
    std::vector<DiagSrc> diag_src;      // dmrgx_kron_diag: where the diagonals of the operator copies sit in the arena
    std::vector<DiagSeg> diag_segs;     // this rank's panels of every KronBlock
    int32_t diag_terms = 0, diag_rounds = 0;
    int64_t diag_NL = 0, diag_NR = 0;
    bool timing = false;                // per-stage HIP-event timing (dmrgx_kron_plan_timing)
    std::vector<hipEvent_t> ev;         // 3 events per recorded apply
    size_t ev_used = 0;
    ~dmrgx_kron_plan() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};

namespace {

dmrgx_status normalise_op(const dmrgx_secop* op, const dmrgx_sectors& sec, const char* what, std::vector<NCell>& out)
{
    out.clear();
    if (!op) return DMRGX_OK;
    if (op->ncells < 0 || (op->ncells > 0 && !op->cells)) DMRGX_FAIL(DMRGX_ERR_ARG, "%s: bad cell list", what);
    for (int32_t i = 0; i < op->ncells; ++i) {
        const dmrgx_cell& c = op->cells[i];
        NCell n;
        if (!op->transposed) { n.q = c.row_sector; n.r0 = c.r0; n.c0 = c.c0; n.nr = c.nr; n.nc = c.nc; n.tr = false; }
        else { n.q = c.row_sector - op->shift; n.r0 = c.c0; n.c0 = c.r0; n.nr = c.nc; n.nc = c.nr; n.tr = true; }
        n.kind = c.kind; n.scale = c.scale; n.data = c.data; n.ld = c.ld;
        const int32_t qc = n.q + op->shift;
        if (n.q < 0 || n.q >= sec.nsec || qc < 0 || qc >= sec.nsec)
            DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "%s: cell %d sector (%d -> %d) out of range [0,%d)", what, i, n.q, qc, sec.nsec);
        if (n.nr <= 0 || n.nc <= 0 || n.r0 < 0 || n.c0 < 0 || n.r0 + n.nr > sec.size[n.q] || n.c0 + n.nc > sec.size[qc])
            DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "%s: cell %d rectangle [%d+%d, %d+%d) exceeds block %d x %d", what, i,
                       n.r0, n.nr, n.c0, n.nc, sec.size[n.q], sec.size[qc]);
        if (n.kind == DMRGX_CELL_IDENT) { if (n.nr != n.nc) DMRGX_FAIL(DMRGX_ERR_ARG, "%s: identity cell %d not square", what, i); }
        else if (n.kind == DMRGX_CELL_DENSE) {
            if (!n.data || n.ld < (n.tr ? n.nr : n.nc)) DMRGX_FAIL(DMRGX_ERR_ARG, "%s: dense cell %d has no data / bad ld", what, i);
        } else DMRGX_FAIL(DMRGX_ERR_ARG, "%s: cell %d has unknown kind %d", what, i, n.kind);
        out.push_back(n);
    }
    return DMRGX_OK;
}

struct Builder {
    std::vector<RelProd> prods;
    std::vector<RelGroup> groups;
    std::vector<GTile> tiles1, tiles2, tiles1b, tiles2b;   // 64x64 and 128x128 lists per stage
    std::vector<int32_t> stage2_groups;
    double flops_alg = 0, flops_exec = 0, flops_alg_big = 0;
    static constexpr bool big = false;      // 128 x 128 tiles for the 128-aligned cores + 64 x 64 remainders in launches of their own measured equal to 64 x 64 alone (rounds 2-4)
    void append_tiles(std::vector<GTile>& b, std::vector<GTile>& s64, int32_t g, int32_t M, int32_t N, int32_t cost) { ggemm_append_tiles_mixed(b, s64, g, M, N, cost, big); }
    int32_t max_split = 1;

    int32_t ksteps(int32_t g) const {      // cost of one tile of group g in k-steps of the GEMM stream
        int32_t c = 0;
        for (int32_t p = groups[g].prod_begin; p < groups[g].prod_end; ++p)
            c += prods[p].kind == GPROD_GEMM ? (prods[p].K + GG_BK - 1) / GG_BK : 1;
        return c;
    }

    // Stage-2 output tiles carry the whole operator list of a KronBlock (K ~ 10^4 at m = 2048) and there are fewer
    // of them than workgroup slots, so long product lists are cut on product boundaries into contiguous segments of
    // about equal length ("split-K"): enough (tile, segment) units that the scheduler can balance the XCDs, long enough
    // that the 32 KB partial-tile write is amortised.  Segment 0 writes y, segment s >= 1 writes a compact slab in the
    // arena; slab_reduce_kernel adds them in fixed order, so the result stays bit-reproducible (no atomics).
    std::vector<RedTask> red_tasks;
    std::vector<RedTile> red_tiles;
    int64_t slab_elems = 0;
    void finalize_stage2(int64_t slab_base) {
        double total = 0;
        for (int32_t g : stage2_groups) {
            const RelGroup& G = groups[g];
            total += (double)ksteps(g) * ((G.M + GG_BM - 1) / GG_BM) * ((G.N + GG_BN - 1) / GG_BN);
        }
        constexpr double units = 8192.0;      // (4 k - 32 k units, a floor of 8 - 32 k-steps and a tapered last segment all measured within +-1 %: round 2)
        // Shortest segment worth its 32 KB slab: 16 k-steps when the launch has work for every workgroup slot anyway; a small
        // superblock (m <= 512: a few thousand tile-k-steps in all) is latency-bound by its longest segment instead, so the
        // floor drops until about 1024 units exist (4 k-steps at least).
        const double min_seg = std::min(16.0, std::max(4.0, total / 1024.0));
        const double seg_target = std::max(total / units, min_seg);
        const size_t ng = stage2_groups.size();
        for (size_t gi = 0; gi < ng; ++gi) {
            const int32_t g = stage2_groups[gi];
            const int32_t cost = ksteps(g);
            const int32_t axpy_begin = groups[g].prod_begin, gemm_begin = groups[g].prod_begin + groups[g].n_axpy, gemm_end = groups[g].prod_end;
            int32_t gcost = 0;
            for (int32_t p = gemm_begin; p < gemm_end; ++p) gcost += (prods[p].K + GG_BK - 1) / GG_BK;
            // Cut points of the group's GEMM stream, in k-steps: equal segments of ~seg_target.  Cuts may fall inside a
            // product: a product is just (pointers, K), so it is split at a multiple of GG_BK.  (Cutting the last
            // segment further into 1/2, 1/4, 1/4 so that every XCD finishes on short units measured neutral to slightly negative.)
            std::vector<int32_t> cuts;
            {
                int32_t S = (int32_t)std::min<double>(64.0, std::max(1.0, std::floor(gcost / seg_target + 0.5)));
                if ((int64_t)groups[g].M * groups[g].N >= (int64_t)1 << 31) S = 1;      // slab_reduce_kernel indexes a block with 32 bits
                if (gcost < 2) S = 1;
                S = std::min(S, gcost);
                for (int32_t i = 1; i < S; ++i) cuts.push_back((int32_t)(((int64_t)gcost * i) / S));
                cuts.push_back(gcost);
                cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
            }
            const int32_t S = (int32_t)cuts.size();
            if (S == 1) { append_tiles(tiles2b, tiles2, g, groups[g].M, groups[g].N, cost); continue; }
            max_split = std::max(max_split, S);
            const int64_t mn = (int64_t)groups[g].M * groups[g].N;
            red_tasks.push_back(RedTask{groups[g].c_off, slab_base + slab_elems, groups[g].ldc, groups[g].M, groups[g].N, S - 1});
            for (int64_t c = 0; c * RED_CHUNK < mn; ++c) red_tiles.push_back(RedTile{(int32_t)red_tasks.size() - 1, (int32_t)c});
            // product start positions in k-steps
            std::vector<int32_t> pstart;
            { int32_t acc = 0; for (int32_t p = gemm_begin; p < gemm_end; ++p) { pstart.push_back(acc); acc += (prods[p].K + GG_BK - 1) / GG_BK; } }
            const int32_t n_axpy = groups[g].n_axpy;
            int32_t lo = 0, pcur = gemm_begin;
            for (int32_t sidx = 0; sidx < S; ++sidx) {
                const int32_t hi = cuts[sidx];
                const int32_t nb = (int32_t)prods.size();
                // the scaled-copy products (identity operator cells, 1 (x) H_R) are dealt over the segments: each is a dependent
                // descriptor + tile load of its own, and all of them on segment 0 made that unit the tail of its tile
                int32_t n_axpy_seg = 0;
                for (int32_t q = sidx; q < n_axpy; q += S) { const RelProd ax = prods[axpy_begin + q]; prods.push_back(ax); ++n_axpy_seg; }
                while (pcur < gemm_end && pstart[pcur - gemm_begin] + (prods[pcur].K + GG_BK - 1) / GG_BK <= lo) ++pcur;   // products are in stream order
                for (int32_t p = pcur; p < gemm_end; ++p) {
                    const int32_t ps = pstart[p - gemm_begin], pe = ps + (prods[p].K + GG_BK - 1) / GG_BK;
                    if (ps >= hi) break;
                    const int32_t olo = std::max(lo, ps), ohi = std::min(hi, pe);
                    if (olo >= ohi) continue;
                    RelProd sub = prods[p];
                    const int32_t k0 = (olo - ps) * GG_BK, k1 = std::min(prods[p].K, (ohi - ps) * GG_BK);
                    sub.a_off += k0; sub.b_off += (int64_t)k0 * sub.ldb; sub.K = k1 - k0;
                    prods.push_back(sub);
                }
                const int32_t ne = (int32_t)prods.size();
                if (sidx == 0) {
                    groups[g].prod_begin = nb; groups[g].prod_end = ne; groups[g].n_axpy = n_axpy_seg;
                    append_tiles(tiles2b, tiles2, g, groups[g].M, groups[g].N, (hi - lo) + n_axpy_seg);
                } else {
                    RelGroup ng2 = groups[g];
                    ng2.c_base = BASE_ARENA;
                    ng2.c_off = slab_base + slab_elems + (int64_t)(sidx - 1) * mn;
                    ng2.ldc = ng2.N;
                    ng2.prod_begin = nb; ng2.prod_end = ne; ng2.n_axpy = n_axpy_seg; ng2.accumulate = 0;
                    groups.push_back(ng2);
                    append_tiles(tiles2b, tiles2, (int32_t)groups.size() - 1, ng2.M, ng2.N, (hi - lo) + n_axpy_seg);
                }
                lo = hi;
            }
            slab_elems += (int64_t)(S - 1) * mn;
        }
    }

    // open a group; products are appended afterwards with add_*; close() sorts AXPY first
    int32_t open(int32_t c_base, int64_t c_off, int32_t ldc, int32_t M, int32_t N, int32_t accumulate) {
        RelGroup g{c_off, c_base, ldc, M, N, (int32_t)prods.size(), (int32_t)prods.size(), 0, accumulate};
        groups.push_back(g);
        return (int32_t)groups.size() - 1;
    }
    void add_gemm(int32_t a_base, int64_t a_off, int32_t lda, int32_t b_base, int64_t b_off, int32_t ldb, int32_t K) {
        prods.push_back(RelProd{a_off, b_off, lda, ldb, K, GPROD_GEMM, 1.0, a_base, b_base});
    }
    void add_axpy(int32_t s_base, int64_t s_off, int32_t lds, double alpha) {
        prods.push_back(RelProd{0, s_off, 0, lds, 0, GPROD_AXPY, alpha, BASE_ARENA, s_base});
    }
    void close(int32_t g, int stage) {
        RelGroup& G = groups[g];
        G.prod_end = (int32_t)prods.size();
        std::stable_sort(prods.begin() + G.prod_begin, prods.end(), [](const RelProd& a, const RelProd& b) { return a.kind > b.kind; });
        G.n_axpy = 0;
        for (int32_t p = G.prod_begin; p < G.prod_end; ++p) {
            const double mn = (double)G.M * G.N;
            if (prods[p].kind == GPROD_AXPY) { G.n_axpy++; flops_alg += 2.0 * mn; }
            else {
                flops_alg += 2.0 * mn * prods[p].K;
                // MFMA work actually issued: 16 x 16 accumulator blocks that intersect the output, k in units of 4
                const double tm = (G.M + 15) / 16, tn = (G.N + 15) / 16;
                flops_exec += 2.0 * tm * tn * 256.0 * (double)(((prods[p].K + 3) / 4) * 4);
                if (big) flops_alg_big += 2.0 * (double)((G.M / 128) * 128) * (double)((G.N / 128) * 128) * prods[p].K;
            }
        }
        if (stage == 1) append_tiles(tiles1b, tiles1, g, G.M, G.N, ksteps(g));
        else stage2_groups.push_back(g);
    }
};

}  // namespace

extern "C" dmrgx_status dmrgx_kron_plan_create(const dmrgx_kron_desc* d, void* stream, dmrgx_kron_plan** out)
{
    hipStream_t st = (hipStream_t)stream;
    if (!d || !out) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_create: null argument");
    *out = nullptr;
    const int32_t W = d->world_size <= 0 ? 1 : d->world_size, me = d->rank;
    if (me < 0 || me >= W) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_create: rank %d outside world %d", me, W);
    const dmrgx_sectors& SL = d->left;
    const dmrgx_sectors& SR = d->right;
    if (SL.nsec <= 0 || SR.nsec <= 0 || !SL.size || !SR.size) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_create: empty sector table");
    for (int i = 0; i < SL.nsec; ++i) if (SL.size[i] <= 0) DMRGX_FAIL(DMRGX_ERR_ARG, "left sector %d has size %d", i, SL.size[i]);
    for (int i = 0; i < SR.nsec; ++i) if (SR.size[i] <= 0) DMRGX_FAIL(DMRGX_ERR_ARG, "right sector %d has size %d", i, SR.size[i]);
    const int32_t nb = d->nblocks;
    if (nb <= 0 || !d->block_il || !d->block_ir) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_create: no KronBlocks");

    // ---- layout ------------------------------------------------------------------------------------------
    std::map<std::pair<int32_t, int32_t>, int32_t> kmap;
    std::vector<int64_t> ref_off(nb + 1, 0);
    for (int32_t k = 0; k < nb; ++k) {
        const int32_t il = d->block_il[k], ir = d->block_ir[k];
        if (il < 0 || il >= SL.nsec || ir < 0 || ir >= SR.nsec) DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "KronBlock %d = (%d,%d) out of range", k, il, ir);
        if (!kmap.emplace(std::make_pair(il, ir), k).second) DMRGX_FAIL(DMRGX_ERR_ARG, "KronBlock (%d,%d) listed twice", il, ir);
        ref_off[k + 1] = ref_off[k] + (int64_t)SL.size[il] * SR.size[ir];
    }
    const int64_t N = ref_off[nb];
    auto nLk = [&](int32_t k) { return SL.size[d->block_il[k]]; };
    auto nRk = [&](int32_t k) { return SR.size[d->block_ir[k]]; };
    // stripes: columns [cbeg(k,w), cend(k,w)) of block k belong to rank w (stripe (w + k) mod W: see stripe_of_rank)
    auto cbeg = [&](int32_t k, int32_t w) { return stripe_cut(nRk(k), W, stripe_of_rank(W, w, k)); };
    auto cend = [&](int32_t k, int32_t w) { return stripe_cut(nRk(k), W, stripe_of_rank(W, w, k) + 1); };
    std::vector<std::vector<int64_t>> seg_off(W, std::vector<int64_t>(nb + 1, 0));
    int64_t max_seg = 0;
    for (int32_t w = 0; w < W; ++w) {
        for (int32_t k = 0; k < nb; ++k) seg_off[w][k + 1] = seg_off[w][k] + (int64_t)nLk(k) * (cend(k, w) - cbeg(k, w));
        max_seg = std::max(max_seg, seg_off[w][nb]);
    }
    const int64_t seg_stride = (W == 1) ? N : ((max_seg + 63) / 64) * 64;
    auto panel_off = [&](int32_t k, int32_t w) { return (int64_t)w * seg_stride + seg_off[w][k]; };   // in a full vector
    auto panel_ld = [&](int32_t k, int32_t w) { return cend(k, w) - cbeg(k, w); };

    // ---- operators ---------------------------------------------------------------------------------------
    if (d->n_left_ops < 0 || d->n_right_ops < 0 || d->nterms < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "negative count");
    std::vector<std::vector<NCell>> Lops(d->n_left_ops), Rops(d->n_right_ops);
    std::vector<NCell> HL, HR;
    for (int32_t i = 0; i < d->n_left_ops; ++i) DMRGX_CHK(normalise_op(&d->left_ops[i], SL, "left op", Lops[i]));
    for (int32_t i = 0; i < d->n_right_ops; ++i) DMRGX_CHK(normalise_op(&d->right_ops[i], SR, "right op", Rops[i]));
    if (d->h_left) { if (d->h_left->shift != 0) DMRGX_FAIL(DMRGX_ERR_ARG, "H_L must have shift 0"); DMRGX_CHK(normalise_op(d->h_left, SL, "H_L", HL)); }
    if (d->h_right) { if (d->h_right->shift != 0) DMRGX_FAIL(DMRGX_ERR_ARG, "H_R must have shift 0"); DMRGX_CHK(normalise_op(d->h_right, SR, "H_R", HR)); }

    // term groups: merge on the side with MORE distinct operators, keyed by the operator on the other side
    std::set<int32_t> usedL, usedR;
    for (int32_t t = 0; t < d->nterms; ++t) {
        const dmrgx_term& T = d->terms[t];
        if (T.left_op < 0 || T.left_op >= d->n_left_ops || T.right_op < 0 || T.right_op >= d->n_right_ops)
            DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "term %d references operator (%d,%d) out of range", t, T.left_op, T.right_op);
        if (d->left_ops[T.left_op].shift + d->right_ops[T.right_op].shift != 0)
            DMRGX_FAIL(DMRGX_ERR_ARG, "term %d does not conserve Sz (shifts %d,%d)", t, d->left_ops[T.left_op].shift, d->right_ops[T.right_op].shift);
        if (T.a != 0.0) { usedL.insert(T.left_op); usedR.insert(T.right_op); }
    }
    // Term groups = a minimum vertex cover of the bipartite graph (distinct left operators) -- terms -- (distinct right operators):
    // a covered right operator B_j keys the group  (sum_t a_t A_t) (x) B_j  of its terms (left operators merged, the map the
    // reference builds at src/DMRGKron.cpp:955-960), a covered left operator A_i the group  A_i (x) (sum_t a_t B_t).  Merging on
    // one side only costs min(#left, #right) groups; at a cut in the middle of a column of the J1-J2 cylinder that is 9-10 sites
    // per operator type against a cover of 8 (Koenig: maximum matching, alternating paths from the unmatched left vertices).
    std::vector<char> coverL(d->n_left_ops, 0), coverR(d->n_right_ops, 0);
    {
        std::vector<std::vector<int32_t>> adj(d->n_left_ops);
        for (int32_t t = 0; t < d->nterms; ++t) if (d->terms[t].a != 0.0) adj[d->terms[t].left_op].push_back(d->terms[t].right_op);
        std::vector<int32_t> matchR(d->n_right_ops, -1), matchL(d->n_left_ops, -1);
        std::vector<char> seen;
        std::function<bool(int32_t)> augment = [&](int32_t l) -> bool {
            for (int32_t r : adj[l]) {
                if (seen[r]) continue;
                seen[r] = 1;
                if (matchR[r] < 0 || augment(matchR[r])) { matchR[r] = l; matchL[l] = r; return true; }
            }
            return false;
        };
        for (int32_t l : usedL) { seen.assign(d->n_right_ops, 0); augment(l); }
        // Z = vertices reachable from unmatched left vertices along alternating paths; cover = (L \ Z) u (R n Z)
        std::vector<char> zL(d->n_left_ops, 0), zR(d->n_right_ops, 0);
        std::vector<int32_t> stack;
        for (int32_t l : usedL) if (matchL[l] < 0) { zL[l] = 1; stack.push_back(l); }
        while (!stack.empty()) {
            const int32_t l = stack.back(); stack.pop_back();
            for (int32_t r : adj[l]) {
                if (zR[r] || matchL[l] == r) continue;
                zR[r] = 1;
                const int32_t l2 = matchR[r];
                if (l2 >= 0 && !zL[l2]) { zL[l2] = 1; stack.push_back(l2); }
            }
        }
        for (int32_t l : usedL) coverL[l] = !zL[l];
        for (int32_t r : usedR) coverR[r] = zR[r];
    }
    struct Group { int32_t sA, sB; std::vector<PCell> left, rightT; };   // rightT: cells of Bhat^T
    std::vector<Group> G;
    std::vector<CopyTask> copies;
    int64_t arena_ops = 0;   // elements
    auto new_dense = [&](int32_t nr, int32_t nc) { int64_t o = arena_ops; arena_ops += (int64_t)nr * nc; return o; };

    // merged (or raw) cell list for one side: sum_t coeff_t * op_t ; transpose_out => store cells transposed
    auto build_side = [&](const std::vector<std::pair<double, const std::vector<NCell>*>>& contrib, bool transpose_out, std::vector<PCell>& dst) {
        std::map<std::tuple<int32_t, int32_t, int32_t, int32_t, int32_t, int32_t>, int32_t> index;   // key -> dst idx
        std::map<int32_t, int32_t> rounds;
        for (auto& ct : contrib) {
            for (const NCell& c : *ct.second) {
                auto key = std::make_tuple(c.q, c.r0, c.c0, c.nr, c.nc, c.kind);
                auto it = index.find(key);
                int32_t di;
                if (it == index.end()) {
                    PCell pc;
                    pc.kind = c.kind; pc.scale = 0.0; pc.off = -1;
                    if (!transpose_out) { pc.q = c.q; pc.r0 = c.r0; pc.c0 = c.c0; pc.nr = c.nr; pc.nc = c.nc; }
                    else { pc.q = c.q; pc.r0 = c.c0; pc.c0 = c.r0; pc.nr = c.nc; pc.nc = c.nr; }   // q stays the ROW sector of the un-transposed op
                    if (c.kind == DMRGX_CELL_DENSE) pc.off = new_dense(pc.nr, pc.nc);
                    dst.push_back(pc);
                    di = (int32_t)dst.size() - 1;
                    index.emplace(key, di);
                } else di = it->second;
                if (c.kind == DMRGX_CELL_IDENT) dst[di].scale += ct.first * c.scale;
                else {
                    CopyTask k;
                    k.dst_off = dst[di].off; k.src = c.data; k.lds = c.ld;
                    k.nr = dst[di].nr; k.nc = dst[di].nc; k.ldd = dst[di].nc;
                    k.tr = (c.tr != transpose_out) ? 1 : 0;
                    k.a = ct.first; k.round = rounds[di]++;
                    copies.push_back(k);
                }
            }
        }
    };

    {
        std::map<std::pair<int32_t, int32_t>, std::vector<int32_t>> by_key;   // (side: 1 = keyed by right op, 0 = by left op; op) -> term indices
        for (int32_t t = 0; t < d->nterms; ++t) {
            const dmrgx_term& T = d->terms[t];
            if (T.a == 0.0) continue;
            if (coverR[T.right_op]) by_key[{1, T.right_op}].push_back(t);
            else if (coverL[T.left_op]) by_key[{0, T.left_op}].push_back(t);
            else DMRGX_FAIL(DMRGX_ERR_INTERNAL, "kron_plan_create: term %d is not covered", t);
        }
        for (auto& kv : by_key) {
            Group g;
            std::vector<std::pair<double, const std::vector<NCell>*>> cl, cr;
            if (kv.first.first == 1) {
                g.sB = d->right_ops[kv.first.second].shift; g.sA = -g.sB;
                cr.push_back({1.0, &Rops[kv.first.second]});
                for (int32_t t : kv.second) cl.push_back({d->terms[t].a, &Lops[d->terms[t].left_op]});
            } else {
                g.sA = d->left_ops[kv.first.second].shift; g.sB = -g.sA;
                cl.push_back({1.0, &Lops[kv.first.second]});
                for (int32_t t : kv.second) cr.push_back({d->terms[t].a, &Rops[d->terms[t].right_op]});
            }
            build_side(cl, false, g.left);
            build_side(cr, true, g.rightT);
            G.push_back(std::move(g));
        }
    }
    std::vector<PCell> PHL, PHRT;
    build_side({{1.0, &HL}}, false, PHL);
    build_side({{1.0, &HR}}, true, PHRT);

    // ---- intermediates: T_{g,k} (n_L(IL') x my stripe of IR) and T_R,k (n_L x my stripe) ------------------
    int64_t arena_T = 0;
    std::vector<std::vector<int64_t>> Toff(G.size(), std::vector<int64_t>(nb, -1));
    std::vector<std::vector<int32_t>> Ksrc(G.size(), std::vector<int32_t>(nb, -1));
    for (size_t g = 0; g < G.size(); ++g)
        for (int32_t k = 0; k < nb; ++k) {
            auto it = kmap.find({d->block_il[k] + G[g].sA, d->block_ir[k] + G[g].sB});
            if (it == kmap.end()) continue;
            Ksrc[g][k] = it->second;
            Toff[g][k] = arena_ops + arena_T;
            arena_T += (int64_t)nLk(it->second) * panel_ld(k, me);
        }
    std::vector<int64_t> TRoff(nb, -1);
    if (!PHRT.empty())
        for (int32_t k = 0; k < nb; ++k) { TRoff[k] = arena_ops + arena_T; arena_T += (int64_t)nLk(k) * panel_ld(k, me); }

    // ---- task tables ---------------------------------------------------------------------------------------
    Builder B;
    // stage 1:  T[:, cols] = sum over the transposed right cells of block (IR -> IR') that reach those columns of
    //           X_src[:, contraction range of the cell] * cellT.  A merged right operator may hold several cells with the
    //           same output columns (e.g. O (x) 1 cell (2,2) and the new site's identity cell (2,1)), so groups are built
    //           per output-column SEGMENT with a product list -- never one overwriting group per cell.
    std::vector<ZeroRect> zero_rects;
    auto stage1 = [&](const std::vector<PCell>& cellsT, int32_t sB, int32_t k, int32_t ksrc, int64_t toff) {
        (void)sB;
        const int32_t ir = d->block_ir[k], cs = cbeg(k, me), ce = cend(k, me), w = ce - cs;
        if (w <= 0) return;
        const int32_t M = nLk(ksrc);
        std::set<int32_t> cuts = {cs, ce};
        auto clampc = [&](int32_t v) { return std::min(std::max(v, cs), ce); };
        for (const PCell& c : cellsT) {
            if (c.q != ir) continue;
            cuts.insert(clampc(c.c0)); cuts.insert(clampc(c.c0 + c.nc));
            if (c.kind == DMRGX_CELL_IDENT)        // a scaled copy must read ONE source panel: cut at panel borders too
                for (int32_t p = 0; p < W; ++p) { const int32_t sc = cbeg(ksrc, p); if (sc > c.r0 && sc < c.r0 + c.nr) cuts.insert(clampc(c.c0 + (sc - c.r0))); }
        }
        std::vector<int32_t> cv(cuts.begin(), cuts.end());
        for (size_t s = 0; s + 1 < cv.size(); ++s) {
            const int32_t o0 = cv[s], o1 = cv[s + 1];
            if (o0 >= o1) continue;
            bool any = false;
            for (const PCell& c : cellsT) if (c.q == ir && c.c0 <= o0 && c.c0 + c.nc >= o1) { any = true; break; }
            if (!any) { zero_rects.push_back(ZeroRect{toff + (o0 - cs), w, M, o1 - o0, 0}); continue; }      // T is zero there: zero_rects_kernel at plan creation
            const int32_t g = B.open(BASE_ARENA, toff + (o0 - cs), w, M, o1 - o0, 0);
            for (const PCell& c : cellsT) {
                if (c.q != ir || c.c0 > o0 || c.c0 + c.nc < o1) continue;
                if (c.kind == DMRGX_CELL_DENSE) {
                    for (int32_t p = 0; p < W; ++p) {            // contraction index r' in [c.r0, c.r0+c.nr) split over source panels
                        const int32_t k0 = std::max(c.r0, cbeg(ksrc, p)), k1 = std::min(c.r0 + c.nr, cend(ksrc, p));
                        if (k0 >= k1) continue;
                        B.add_gemm(BASE_X, panel_off(ksrc, p) + (k0 - cbeg(ksrc, p)), panel_ld(ksrc, p),
                                   BASE_ARENA, c.off + (int64_t)(k0 - c.r0) * c.nc + (o0 - c.c0), c.nc, k1 - k0);
                    }
                } else {                                         // identity cell: T[:, o] += scale * X_src[:, c.r0 + (o - c.c0)]
                    const int32_t s0 = c.r0 + (o0 - c.c0);
                    int32_t p = 0;
                    for (int32_t pp = 0; pp < W; ++pp) if (cbeg(ksrc, pp) <= s0 && s0 < cend(ksrc, pp)) p = pp;
                    B.add_axpy(BASE_X, panel_off(ksrc, p) + (s0 - cbeg(ksrc, p)), panel_ld(ksrc, p), c.scale);
                }
            }
            B.close(g, 1);
        }
    };
    for (size_t g = 0; g < G.size(); ++g)
        for (int32_t k = 0; k < nb; ++k) if (Ksrc[g][k] >= 0) stage1(G[g].rightT, G[g].sB, k, Ksrc[g][k], Toff[g][k]);
    if (!PHRT.empty()) for (int32_t k = 0; k < nb; ++k) stage1(PHRT, 0, k, k, TRoff[k]);
    const int32_t n_groups_stage1 = (int32_t)B.groups.size();

    // stage 2:  Y_k[rows, stripe] = sum over left cells covering `rows`
    for (int32_t k = 0; k < nb; ++k) {
        const int32_t il = d->block_il[k], w = panel_ld(k, me), nl = nLk(k);
        if (w <= 0) continue;
        std::set<int32_t> cuts = {0, nl};
        for (size_t g = 0; g < G.size(); ++g) if (Ksrc[g][k] >= 0) for (const PCell& c : G[g].left) if (c.q == il) { cuts.insert(c.r0); cuts.insert(c.r0 + c.nr); }
        for (const PCell& c : PHL) if (c.q == il) { cuts.insert(c.r0); cuts.insert(c.r0 + c.nr); }
        std::vector<int32_t> cv(cuts.begin(), cuts.end());
        for (size_t s = 0; s + 1 < cv.size(); ++s) {
            const int32_t ra = cv[s], rb = cv[s + 1];
            const int32_t grp = B.open(BASE_Y, seg_off[me][k] + (int64_t)ra * w, w, rb - ra, w, 0);
            for (size_t g = 0; g < G.size(); ++g) {
                if (Ksrc[g][k] < 0) continue;
                for (const PCell& c : G[g].left) {
                    if (c.q != il || c.r0 > ra || c.r0 + c.nr < rb) continue;
                    if (c.kind == DMRGX_CELL_DENSE)
                        B.add_gemm(BASE_ARENA, c.off + (int64_t)(ra - c.r0) * c.nc, c.nc, BASE_ARENA, Toff[g][k] + (int64_t)c.c0 * w, w, c.nc);
                    else
                        B.add_axpy(BASE_ARENA, Toff[g][k] + (int64_t)(c.c0 + (ra - c.r0)) * w, w, c.scale);
                }
            }
            for (const PCell& c : PHL) {                      // H_L (x) 1 : B operand is this rank's own panel of X_k
                if (c.q != il || c.r0 > ra || c.r0 + c.nr < rb) continue;
                if (c.kind == DMRGX_CELL_DENSE)
                    B.add_gemm(BASE_ARENA, c.off + (int64_t)(ra - c.r0) * c.nc, c.nc, BASE_X, panel_off(k, me) + (int64_t)c.c0 * w, w, c.nc);
                else
                    B.add_axpy(BASE_X, panel_off(k, me) + (int64_t)(c.c0 + (ra - c.r0)) * w, w, c.scale);
            }
            if (TRoff[k] >= 0) B.add_axpy(BASE_ARENA, TRoff[k] + (int64_t)ra * w, w, 1.0);   // 1 (x) H_R
            B.close(grp, 2);
        }
    }

    const int64_t slab_base = arena_ops + arena_T;
    B.finalize_stage2(slab_base);
    const int64_t arena_slabs = B.slab_elems;
    ggemm_schedule(B.tiles1, B.groups);
    ggemm_schedule(B.tiles2, B.groups);
    ggemm_schedule(B.tiles1b, B.groups, 2);
    ggemm_schedule(B.tiles2b, B.groups, 2);
    if (const char* dump = getenv("DMRGX_PLAN_DUMP")) {   // developer aid: scheduled tile lists, one line per tile
        if (FILE* f = fopen(dump, "w")) {
            auto put = [&](const char* name, const std::vector<GTile>& tl) {
                for (size_t i = 0; i < tl.size(); ++i) {
                    const GTile& t = tl[i];
                    if (t.group < 0) { fprintf(f, "%s %zu -1 0 0 0 0 0 0\n", name, i); continue; }
                    const RelGroup& G2 = B.groups[t.group];
                    fprintf(f, "%s %zu %d %d %d %d %d %d %d\n", name, i, t.group, t.tm, t.tn, G2.M, G2.N, B.ksteps(t.group), G2.prod_end - G2.prod_begin);
                }
            };
            put("s1", B.tiles1); put("s2", B.tiles2); put("s1b", B.tiles1b); put("s2b", B.tiles2b);
            fclose(f);
        }
    }

    // ---- device objects ------------------------------------------------------------------------------------
    dmrgx_kron_plan* P = new (std::nothrow) dmrgx_kron_plan();
    if (!P) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
    std::unique_ptr<dmrgx_kron_plan> guard(P);
    P->world = W; P->rank = me;
    DMRGX_CHK(P->arena.alloc((size_t)std::max<int64_t>(arena_ops + arena_T + arena_slabs, 1) * sizeof(double)));
    P->n_red_tiles = (int32_t)B.red_tiles.size();
    {   // operator copies, one launch per accumulation round (round 0 writes); the unreached segments of the intermediates are zeroed.
        // Nothing else of the arena is read before it is written: every dense operator cell has a round-0 copy, stage 1 writes the reached
        // segments of every T_{g,k} whole, the split-K segments write their slabs whole.
        int32_t max_round = -1;
        for (auto& c : copies) max_round = std::max(max_round, c.round);
        DevBuf d_tab;                                   // the copy tasks and the tile lists of all rounds in one upload
        PackedUpload pk;
        const size_t o_tasks = pk.add(copies);
        const size_t o_zero = pk.add(zero_rects);
        std::vector<std::pair<size_t, size_t>> lists;
        for (int32_t r = 0; r <= max_round; ++r) {
            std::vector<CopyTile> ct;
            for (size_t i = 0; i < copies.size(); ++i) if (copies[i].round == r)
                for (int32_t ti = 0; ti < (copies[i].nr + 31) / 32; ++ti)
                    for (int32_t tj = 0; tj < (copies[i].nc + 31) / 32; ++tj) ct.push_back(CopyTile{(int32_t)i, ti, tj, 0});
            if (ct.empty()) continue;
            lists.push_back({pk.add(ct), ct.size()});
        }
        DMRGX_CHK(pk.upload(d_tab, st));
        if (!zero_rects.empty()) {
            int64_t big = 1;
            for (const ZeroRect& r : zero_rects) big = std::max(big, (int64_t)r.nr * r.nc);
            hipLaunchKernelGGL(zero_rects_kernel, dim3((unsigned)std::min<int64_t>((big + 2047) / 2048, 512), (unsigned)zero_rects.size()), dim3(256), 0, st,
                               (const ZeroRect*)packed_at<ZeroRect>(d_tab, o_zero), P->arena.as<double>());
            DMRGX_HIP(hipGetLastError());
        }
        for (const auto& l : lists) {
            hipLaunchKernelGGL(cell_copy_kernel, dim3((unsigned)l.second), dim3(256), 0, st, (const CopyTile*)packed_at<CopyTile>(d_tab, l.first), (const CopyTask*)packed_at<CopyTask>(d_tab, o_tasks), P->arena.as<double>());
            DMRGX_HIP(hipGetLastError());
        }
    }
    P->nprods = (int32_t)B.prods.size(); P->ngroups = (int32_t)B.groups.size();
    P->ntiles1 = (int32_t)B.tiles1.size(); P->ntiles2 = (int32_t)B.tiles2.size();
    P->ntiles1b = (int32_t)B.tiles1b.size(); P->ntiles2b = (int32_t)B.tiles2b.size();
    {   // every table of the plan in one copy; the members are views into P->d_tables
        std::vector<LayoutSeg> segs;                       // layout conversion table (reference order <-> rank-major stripes)
        for (int32_t k = 0; k < nb; ++k) for (int32_t w = 0; w < W; ++w) {
            if (panel_ld(k, w) <= 0) continue;
            segs.push_back(LayoutSeg{ref_off[k] + cbeg(k, w), panel_off(k, w), nLk(k), panel_ld(k, w), nRk(k), panel_ld(k, w)});
        }
        P->nlayout = (int32_t)segs.size();
        PackedUpload pk;
        const size_t o0 = pk.add(B.prods), o1 = pk.add(B.groups), o2 = pk.add(B.tiles1), o3 = pk.add(B.tiles2), o4 = pk.add(B.tiles1b), o5 = pk.add(B.tiles2b),
                     o8 = pk.add(B.red_tasks), o9 = pk.add(B.red_tiles), o10 = pk.add(segs);
        DMRGX_CHK(pk.upload(P->d_tables, st));
        PackedUpload::view<std::decay<decltype(B.prods[0])>::type>(P->d_rprods, P->d_tables, o0, B.prods.size());
        PackedUpload::view<std::decay<decltype(B.groups[0])>::type>(P->d_rgroups, P->d_tables, o1, B.groups.size());
        PackedUpload::view<GTile>(P->d_tiles1, P->d_tables, o2, B.tiles1.size());
        PackedUpload::view<GTile>(P->d_tiles2, P->d_tables, o3, B.tiles2.size());
        PackedUpload::view<GTile>(P->d_tiles1b, P->d_tables, o4, B.tiles1b.size());
        PackedUpload::view<GTile>(P->d_tiles2b, P->d_tables, o5, B.tiles2b.size());
        PackedUpload::view<std::decay<decltype(B.red_tasks[0])>::type>(P->d_red_tasks, P->d_tables, o8, B.red_tasks.size());
        PackedUpload::view<std::decay<decltype(B.red_tiles[0])>::type>(P->d_red_tiles, P->d_tables, o9, B.red_tiles.size());
        PackedUpload::view<LayoutSeg>(P->d_layout, P->d_tables, o10, segs.size());
    }

    {   // diagonal terms (dmrgx_kron_diag): t = 0: H_L (x) 1, t = 1: 1 (x) H_R, then the shift-0 groups
        std::vector<int64_t> offL(SL.nsec + 1, 0), offR(SR.nsec + 1, 0);
        for (int i = 0; i < SL.nsec; ++i) offL[i + 1] = offL[i] + SL.size[i];
        for (int i = 0; i < SR.nsec; ++i) offR[i + 1] = offR[i] + SR.size[i];
        P->diag_NL = offL[SL.nsec]; P->diag_NR = offR[SR.nsec];
        int32_t nt = 2;
        for (auto& g : G) if (g.sA == 0 && g.sB == 0) ++nt;
        P->diag_terms = nt;
        const int64_t baseB = (int64_t)nt * P->diag_NL;                       // dvec = [dA (nt x NL) | dB (nt x NR)]
        std::map<std::tuple<int32_t, int32_t, int32_t>, int32_t> rounds;      // (side, term, sector) -> sources so far
        auto add_cells = [&](const std::vector<PCell>& cells, int side, int32_t t, const std::vector<int64_t>& off, bool transposed_storage) {
            for (const PCell& c : cells) {
                // block (q -> q): the diagonal crosses the cell where r0 + i == c0 + j (transposed storage swaps the roles, same set)
                const int32_t lo = std::max(c.r0, c.c0), hi = std::min(c.r0 + c.nr, c.c0 + c.nc);
                if (lo >= hi) continue;
                if (c.kind != DMRGX_CELL_DENSE && c.r0 != c.c0) continue;      // an identity cell maps row r0 + i to column c0 + i: on the diagonal only when r0 == c0
                DiagSrc d;
                d.n = hi - lo; d.scale = c.scale; d.ld = c.nc;
                d.off = c.kind == DMRGX_CELL_DENSE ? c.off + (int64_t)(lo - c.r0) * c.nc + (lo - c.c0) : -1;
                d.dst = (side == 0 ? (int64_t)t * P->diag_NL : baseB + (int64_t)t * P->diag_NR) + off[c.q] + lo;
                d.round = rounds[std::make_tuple(side, t, c.q)]++;
                P->diag_rounds = std::max(P->diag_rounds, d.round + 1);
                P->diag_src.push_back(d);
                (void)transposed_storage;
            }
        };
        auto add_ones = [&](int side, int32_t t, const dmrgx_sectors& S, const std::vector<int64_t>& off) {
            for (int32_t q = 0; q < S.nsec; ++q) {
                DiagSrc d;
                d.n = S.size[q]; d.scale = 1.0; d.ld = 0; d.off = -1; d.round = rounds[std::make_tuple(side, t, q)]++;
                d.dst = (side == 0 ? (int64_t)t * P->diag_NL : baseB + (int64_t)t * P->diag_NR) + off[q];
                P->diag_rounds = std::max(P->diag_rounds, d.round + 1);
                P->diag_src.push_back(d);
            }
        };
        add_cells(PHL, 0, 0, offL, false); add_ones(1, 0, SR, offR);
        add_ones(0, 1, SL, offL); add_cells(PHRT, 1, 1, offR, true);
        int32_t t = 2;
        for (auto& g : G) {
            if (g.sA != 0 || g.sB != 0) continue;
            add_cells(g.left, 0, t, offL, false); add_cells(g.rightT, 1, t, offR, true);
            ++t;
        }
        for (int32_t k = 0; k < nb; ++k) {
            if (panel_ld(k, me) <= 0) continue;
            P->diag_segs.push_back(DiagSeg{seg_off[me][k], nLk(k), panel_ld(k, me), offL[d->block_il[k]], offR[d->block_ir[k]] + cbeg(k, me)});
        }
    }

    dmrgx_kron_info& I = P->info;
    I.n_states = N; I.vec_len = (W == 1) ? N : (int64_t)W * seg_stride; I.local_offset = (int64_t)me * seg_stride;
    I.local_len = (W == 1) ? N : seg_stride; I.seg_stride = seg_stride;
    I.flops_alg = B.flops_alg; I.flops_exec = B.flops_exec;
    double opbytes = 0;
    auto cellbytes = [&](const std::vector<PCell>& v) { for (auto& c : v) if (c.kind == DMRGX_CELL_DENSE) opbytes += 8.0 * c.nr * c.nc; };
    for (auto& g : G) { cellbytes(g.left); cellbytes(g.rightT); }
    cellbytes(PHL); cellbytes(PHRT);
    I.bytes_alg = opbytes + 8.0 * ((double)N + (double)seg_off[me][nb]);
    I.bytes_workspace = 16.0 * (double)arena_T + 16.0 * (double)B.slab_elems;
    I.n_groups = (int32_t)G.size(); I.n_tiles_stage1 = P->ntiles1 + P->ntiles1b; I.n_tiles_stage2 = P->ntiles2 + P->ntiles2b;
    I.n_tiles_big = P->ntiles1b + P->ntiles2b; I.flops_alg_big = B.flops_alg_big;
    (void)n_groups_stage1;
    *out = guard.release();
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_plan_info(const dmrgx_kron_plan* plan, dmrgx_kron_info* info)
{
    if (!plan || !info) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_info: null argument");
    *info = plan->info;
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_apply(dmrgx_kron_plan* P, const double* x_full, double* y_local, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!P || !x_full || !y_local) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_apply: null argument");
    dmrgx_kron_plan::Patched* T = nullptr;
    for (auto& c : P->patched) if (c->x == x_full && c->y == y_local) { T = c.get(); break; }
    if (!T) {
        if (P->patched.size() < dmrgx_kron_plan::PATCHED_MAX) {
            std::unique_ptr<dmrgx_kron_plan::Patched> c(new (std::nothrow) dmrgx_kron_plan::Patched());
            if (!c) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
            DMRGX_CHK(c->prods.alloc(std::max<size_t>((size_t)P->nprods, 1) * sizeof(GProd)));
            DMRGX_CHK(c->groups.alloc(std::max<size_t>((size_t)P->ngroups, 1) * sizeof(GGroup)));
            P->patched.push_back(std::move(c));
            T = P->patched.back().get();
        } else {
            T = P->patched[P->patched_next].get();
            P->patched_next = (P->patched_next + 1) % dmrgx_kron_plan::PATCHED_MAX;
        }
        T->x = x_full; T->y = y_local;
        const int n = std::max(P->nprods, P->ngroups);
        if (n > 0) {
            hipLaunchKernelGGL(patch_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                               P->d_rprods.as<RelProd>(), T->prods.as<GProd>(), P->nprods,
                               P->d_rgroups.as<RelGroup>(), T->groups.as<GGroup>(), P->ngroups,
                               P->arena.as<double>(), x_full, y_local);
            DMRGX_HIP(hipGetLastError());
        }
    }
    hipEvent_t* e = nullptr;
    if (P->timing && P->ev_used + 5 <= 5 * 4096) {
        while (P->ev.size() < P->ev_used + 5) { hipEvent_t x; DMRGX_HIP(hipEventCreate(&x)); P->ev.push_back(x); }
        e = &P->ev[P->ev_used];
        P->ev_used += 5;
    }
    // four launches: {stage 1, stage 2} x {128x128 core tiles, 64x64 remainder tiles}, each bracketed by events
    if (e) DMRGX_HIP(hipEventRecord(e[0], st));
    DMRGX_CHK(ggemm_launch(P->d_tiles1b.as<GTile>(), T->groups.as<GGroup>(), T->prods.as<GProd>(), P->ntiles1b, st, 1));
    if (e) DMRGX_HIP(hipEventRecord(e[1], st));
    DMRGX_CHK(ggemm_launch(P->d_tiles1.as<GTile>(), T->groups.as<GGroup>(), T->prods.as<GProd>(), P->ntiles1, st, 0));
    if (e) DMRGX_HIP(hipEventRecord(e[2], st));
    DMRGX_CHK(ggemm_launch(P->d_tiles2b.as<GTile>(), T->groups.as<GGroup>(), T->prods.as<GProd>(), P->ntiles2b, st, 1));
    if (e) DMRGX_HIP(hipEventRecord(e[3], st));
    DMRGX_CHK(ggemm_launch(P->d_tiles2.as<GTile>(), T->groups.as<GGroup>(), T->prods.as<GProd>(), P->ntiles2, st, 0));
    if (e) DMRGX_HIP(hipEventRecord(e[4], st));
    if (P->n_red_tiles > 0) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)P->n_red_tiles), dim3(256), 0, st, P->d_red_tiles.as<RedTile>(), P->d_red_tasks.as<RedTask>(), y_local, P->arena.as<double>());
        DMRGX_HIP(hipGetLastError());
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_plan_timing(dmrgx_kron_plan* P, int32_t enable)
{
    if (!P) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_timing: null plan");
    if (enable) P->ev_used = 0;
    P->timing = enable != 0;
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_plan_timing_read(dmrgx_kron_plan* P, double* ms, int64_t* n_applies)
{
    if (!P || !ms || !n_applies) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_plan_timing_read: null argument");
    ms[0] = ms[1] = ms[2] = ms[3] = 0.0; *n_applies = (int64_t)(P->ev_used / 5);
    for (size_t i = 0; i + 4 < P->ev_used; i += 5) {
        DMRGX_HIP(hipEventSynchronize(P->ev[i + 4]));
        for (int k = 0; k < 4; ++k) { float t = 0; DMRGX_HIP(hipEventElapsedTime(&t, P->ev[i + k], P->ev[i + k + 1])); ms[k] += t; }
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_plan_destroy(dmrgx_kron_plan* plan)
{
    if (!plan) return DMRGX_OK;
    // no synchronisation: the blocks go back to the pool and are recycled in stream order (pool.hip)
    delete plan;
    return DMRGX_OK;
}

static dmrgx_status layout_copy(const dmrgx_kron_plan* P, const double* src, double* dst, int to_striped, hipStream_t st)
{
    if (!P || !src || !dst) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_vec layout copy: null argument");
    if (P->nlayout == 0) return DMRGX_OK;
    hipLaunchKernelGGL(layout_copy_kernel, dim3(64, (unsigned)P->nlayout), dim3(256), 0, st, P->d_layout.as<LayoutSeg>(), src, dst, to_striped);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_kron_vec_to_striped(const dmrgx_kron_plan* plan, const double* v_ref_dev, double* v_full_dev, void* stream)
{ return layout_copy(plan, v_ref_dev, v_full_dev, 1, (hipStream_t)stream); }

extern "C" dmrgx_status dmrgx_kron_vec_from_striped(const dmrgx_kron_plan* plan, const double* v_full_dev, double* v_ref_dev, void* stream)
{ return layout_copy(plan, v_full_dev, v_ref_dev, 0, (hipStream_t)stream); }

extern "C" dmrgx_status dmrgx_kron_diag(dmrgx_kron_plan* P, double* d_local, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!P || !d_local) DMRGX_FAIL(DMRGX_ERR_ARG, "kron_diag: null argument");
    DevBuf dvec, d_src, d_segs;
    const size_t nvec = (size_t)P->diag_terms * (size_t)(P->diag_NL + P->diag_NR);
    DMRGX_CHK(dvec.alloc(std::max<size_t>(nvec, 1) * sizeof(double)));
    DMRGX_HIP(zero_async(dvec.p, dvec.bytes, st));
    DMRGX_HIP(zero_async(d_local, (size_t)P->info.local_len * sizeof(double), st));
    DMRGX_CHK(upload(d_src, P->diag_src, st));
    DMRGX_CHK(upload(d_segs, P->diag_segs, st));
    for (int32_t r = 0; r < P->diag_rounds; ++r) {
        hipLaunchKernelGGL(diag_gather_kernel, dim3((unsigned)P->diag_src.size()), dim3(256), 0, st, d_src.as<DiagSrc>(), (int)P->diag_src.size(), r,
                           (const double*)P->arena.as<double>(), dvec.as<double>());
        DMRGX_HIP(hipGetLastError());
    }
    if (!P->diag_segs.empty()) {
        hipLaunchKernelGGL(diag_fill_kernel, dim3(64, (unsigned)P->diag_segs.size()), dim3(256), 0, st, d_segs.as<DiagSeg>(), (const double*)dvec.as<double>(),
                           (const double*)(dvec.as<double>() + (size_t)P->diag_terms * P->diag_NL), P->diag_terms, P->diag_NL, P->diag_NR, d_local);
        DMRGX_HIP(hipGetLastError());
    }
    return DMRGX_OK;
}
