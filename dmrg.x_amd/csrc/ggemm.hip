// Grouped f64 GEMM for gfx950 (MI355X): 64x64 output tile per 256-thread workgroup, 4 waves in a 2x2
// arrangement, each wave owning a 32x32 sub-tile as 2x2 v_mfma_f64_16x16x4_f64 accumulators.
//
// Operand staging (HBM -> registers -> LDS, one barrier per 16-deep k-step, register prefetch of the next
// k-step while the current one is multiplied):
//   A tile 64 x 16, stored [i][k] with row stride 17 doubles  -> the MFMA A fragment (lane: i=l&15, k=l>>4)
//                                                                 reads all 64 LDS banks exactly once
//   B tile 16 x 64, stored [k][j] with row stride 80 doubles  -> same for the B fragment (k=l>>4, j=l&15)
// C/D fragment of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg  (verified on hardware,
// profiles/r01_mfma_f64_probe.txt).
//
// The k-steps of all GEMM products of a group form one software-pipelined stream, so a group made of many
// short products (the superblock MatMult's stage 2: up to ~26 operator cells per output tile) keeps the
// pipeline full across product boundaries.
#include "ggemm.h"
#include <algorithm>
#include <cstdlib>

namespace dmrgx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1)))* gptr;   // global-memory pointer (see GG_GLOAD)
typedef double __attribute__((address_space(1)))* gwptr;
typedef const char __attribute__((address_space(1)))* gbptr;    // byte pointer for base + 32-bit offset addressing

// LDS strides are derived inside the kernel template: A rows 17 doubles apart (34 dwords: the 16 rows of a fragment
// hit 16 distinct 2-bank slots under ds_read_b64's 64 banks and under the ds_read2_b64 the compiler fuses k-steps into:
// 32 banks, 16-lane groups; 18 was 2-way conflicted there), B rows BN+16 doubles apart.

// Workgroup -> tile: tiles[blockIdx.x].  The host (ggemm_schedule) lays the list out so that entries b, b+8,
// b+16, ... -- the blocks the dispatcher deals to one XCD -- form that XCD's cost-balanced, locality-clustered
// work list; entries with group < 0 are padding.

// TR x TC v_mfma_f64_16x16x4 accumulators per wave, WR x WC waves per workgroup: tile (16 TR WR) x (16 TC WC).
//   <2,2,2,2>:  64 x  64 tile, 256 threads, 4 workgroups per CU -- ragged remainders and small sectors
//   <4,2,2,4>: 128 x 128 tile, 512 threads, 2 workgroups per CU -- same 4 waves per SIMD, half the L2->LDS bytes
//                                                                 per flop, 64 accumulator registers per wave
template <int TR, int TC, int WR, int WC>
__global__ void __launch_bounds__(64 * WR * WC, 4)
ggemm_kernel(const GTile* __restrict__ tiles, const GGroup* __restrict__ groups, const GProd* __restrict__ prods, int ntiles)
{
    constexpr int THREADS = 64 * WR * WC;
    constexpr int BM = 16 * TR * WR, BN = 16 * TC * WC, BK = GG_BK;
    constexpr int AS_LD = BK + 1;        // 17: see header comment (conflict-free under ds_read_b64 and ds_read2_b64)
    constexpr int BS_LD = BN + 16;       // == 32 dwords (mod 64): rows k and k+1 use disjoint bank halves
    constexpr int AROWS = THREADS / 16;  // A rows covered per pass
    constexpr int NA = BM / AROWS;       // A elements per thread per k-step
    constexpr int BROWS = THREADS / BN;  // B rows covered per pass
    constexpr int NB = BK / BROWS;       // B elements per thread per k-step
    __shared__ double As[2][BM * AS_LD];
    __shared__ double Bs[2][BK * BS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GTile tile = tiles[blockIdx.x];
    if (tile.group < 0) return;
    const GGroup g = groups[tile.group];
    const int m0 = tile.tm * GG_BM, n0 = tile.tn * GG_BN;        // tile coordinates are in 64-units for both shapes
    const int mrem = min(BM, g.M - m0), nrem = min(BN, g.N - n0);
    const bool full_mn = (mrem == BM) && (nrem == BN);
    const int wr = wave / WC, wc = wave % WC;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int wrow = wr * 16 * TR, wcol = wc * 16 * TC;           // wave sub-tile origin

    d4 acc[TR][TC];
#pragma unroll
    for (int mi = 0; mi < TR; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

    // ---- scaled-copy products (identity operator cells): acc += alpha * S[tile] ------------------------
    int p = g.prod_begin;
    for (int e = p + g.n_axpy; p < e; ++p) {
        const GProd pr = prods[p];
        gptr S = (gptr)(pr.B + (size_t)m0 * pr.ldb + n0);
#pragma unroll
        for (int mi = 0; mi < TR; ++mi)
#pragma unroll
            for (int ni = 0; ni < TC; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wrow + mi * 16 + l4 + 4 * r, col = wcol + ni * 16 + l15;
                    if (row < mrem && col < nrem) acc[mi][ni][r] += pr.alpha * S[(size_t)row * pr.ldb + col];
                }
    }

    // ---- GEMM stream ------------------------------------------------------------------------------------
    const int a_r = tid >> 4, a_k = tid & 15;      // A loader: rows a_r + AROWS s, column a_k (128 B per 16 lanes)
    const int b_k = tid / BN, b_j = tid % BN;      // B loader: rows b_k + BROWS s, column b_j (512 B per wave)
    double ra[NA], rb[NB];
    const int pend = g.prod_end;
    int k0 = 0;
    const double *cA = nullptr, *cB = nullptr;     // current product, wave-uniform -> SGPRs
    int clda = 0, cldb = 0, cK = 0;

// Operand pointers come out of the task table, so the compiler would treat them as generic and emit flat_load
// (+ lgkmcnt waits that serialise against LDS); they are global by construction -> explicit address space.
// Interior k-steps (full tile, full BK) take a uniform fast path with plain loads; edge k-steps load from clamped,
// always-valid addresses and zero the out-of-range elements by a 0/1 multiply (a select would let the compiler sink
// each load under its own exec-mask branch and wait on it individually).
#define GG_GLOAD(KK0)                                                                         \
    {                                                                                         \
        gbptr A_ = (gbptr)(cA + (size_t)m0 * clda + (KK0));    /* wave-uniform byte bases: saddr + 32-bit voffset loads */ \
        gbptr B_ = (gbptr)(cB + (size_t)(KK0) * cldb + n0);                                   \
        const int klast_ = cK - 1 - (KK0);                                                    \
        if (full_mn && klast_ >= BK - 1) {                                                    \
            unsigned ao_ = ((unsigned)a_r * (unsigned)clda + (unsigned)a_k) * 8u, bo_ = ((unsigned)b_k * (unsigned)cldb + (unsigned)b_j) * 8u; \
            const unsigned as_ = (unsigned)(8 * AROWS) * (unsigned)clda, bs_ = (unsigned)(8 * BROWS) * (unsigned)cldb; \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) { ra[s] = *(gptr)(A_ + ao_); ao_ += as_; } \
            _Pragma("unroll") for (int s = 0; s < NB; ++s) { rb[s] = *(gptr)(B_ + bo_); bo_ += bs_; } \
        } else {                                                                              \
            const double ak_ = a_k <= klast_ ? 1.0 : 0.0, bj_ = b_j < nrem ? 1.0 : 0.0;       \
            const unsigned acol_ = (unsigned)min(a_k, klast_), bcol_ = (unsigned)min(b_j, nrem - 1); \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) {                                  \
                const int row_ = a_r + AROWS * s;                                             \
                ra[s] = *(gptr)(A_ + ((unsigned)min(row_, mrem - 1) * (unsigned)clda + acol_) * 8u) * (row_ < mrem ? ak_ : 0.0); \
            }                                                                                 \
            _Pragma("unroll") for (int s = 0; s < NB; ++s) {                                  \
                const int k_ = b_k + BROWS * s;                                               \
                rb[s] = *(gptr)(B_ + ((unsigned)min(k_, klast_) * (unsigned)cldb + bcol_) * 8u) * (k_ <= klast_ ? bj_ : 0.0); \
            }                                                                                 \
        }                                                                                     \
    }
#define GG_LSTORE(BUF)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) As[BUF][(a_r + AROWS * s) * AS_LD + a_k] = ra[s]; \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) Bs[BUF][(b_k + BROWS * s) * BS_LD + b_j] = rb[s]; \
    }

    bool have = p < pend;
    if (have) {
        cA = prods[p].A; cB = prods[p].B; clda = prods[p].lda; cldb = prods[p].ldb; cK = prods[p].K;
        GG_GLOAD(0);
        GG_LSTORE(0);
    }
    __syncthreads();
    int buf = 0;
    while (have) {
        int pn = p, kn = k0 + BK;
        if (kn >= cK) { pn = p + 1; kn = 0; }
        const bool have_next = pn < pend;
        if (have_next) {
            if (pn != p) { cA = prods[pn].A; cB = prods[pn].B; clda = prods[pn].lda; cldb = prods[pn].ldb; cK = prods[pn].K; }
            GG_GLOAD(kn);
        }
        const double* as = &As[buf][(wrow + l15) * AS_LD + l4];
        const double* bs = &Bs[buf][l4 * BS_LD + wcol + l15];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double a[TR], b[TC];
#pragma unroll
            for (int mi = 0; mi < TR; ++mi) a[mi] = as[mi * 16 * AS_LD + kk];
#pragma unroll
            for (int ni = 0; ni < TC; ++ni) b[ni] = bs[kk * BS_LD + ni * 16];
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (have_next) GG_LSTORE(buf ^ 1);
        __syncthreads();
        buf ^= 1; p = pn; k0 = kn; have = have_next;
    }
#undef GG_GLOAD
#undef GG_LSTORE

    // ---- epilogue ---------------------------------------------------------------------------------------
    gwptr C = (gwptr)(g.C + (size_t)m0 * g.ldc + n0);
#pragma unroll
    for (int mi = 0; mi < TR; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wrow + mi * 16 + l4 + 4 * r, col = wcol + ni * 16 + l15;
                if (row < mrem && col < nrem) {
                    gwptr c = C + (size_t)row * g.ldc + col;
                    *c = g.accumulate ? (*c + acc[mi][ni][r]) : acc[mi][ni][r];
                }
            }
}

bool ggemm_use_big_tiles()
{
    const char* e = getenv("DMRGX_TILES");
    return e && std::string(e) == "mixed";   // the plan's ragged task tables currently balance better on 64 x 64 tiles only
}

void ggemm_schedule(std::vector<GTile>& tiles)
{
    constexpr int NX = 8, CLUSTER = 8;
    if (tiles.empty()) return;
    struct Cl { size_t begin, end; int64_t cost; };
    std::vector<Cl> cl;
    for (size_t i = 0; i < tiles.size();) {
        size_t j = i;
        int64_t c = 0;
        while (j < tiles.size() && j - i < (size_t)CLUSTER && tiles[j].group == tiles[i].group && tiles[j].tm == tiles[i].tm) { c += tiles[j].pad + 2; ++j; }
        cl.push_back(Cl{i, j, c});
        i = j;
    }
    std::stable_sort(cl.begin(), cl.end(), [](const Cl& a, const Cl& b) { return a.cost > b.cost; });
    std::vector<std::vector<GTile>> bins(NX);
    int64_t load[NX] = {0};
    for (const Cl& c : cl) {
        int best = 0;
        for (int x = 1; x < NX; ++x) if (load[x] < load[best]) best = x;
        load[best] += c.cost;
        for (size_t t = c.begin; t < c.end; ++t) bins[best].push_back(tiles[t]);
    }
    size_t len = 0;
    for (auto& b : bins) len = std::max(len, b.size());
    std::vector<GTile> out(len * NX, GTile{-1, 0, 0, 0});
    for (int x = 0; x < NX; ++x) for (size_t i = 0; i < bins[x].size(); ++i) out[i * NX + x] = bins[x][i];
    tiles.swap(out);
}

dmrgx_status ggemm_launch(const GTile* d_tiles, const GGroup* d_groups, const GProd* d_prods, int32_t ntiles, hipStream_t st, int big)
{
    if (ntiles <= 0) return DMRGX_OK;
    if (big) hipLaunchKernelGGL((ggemm_kernel<4, 2, 2, 4>), dim3((unsigned)ntiles), dim3(512), 0, st, d_tiles, d_groups, d_prods, ntiles);
    else hipLaunchKernelGGL((ggemm_kernel<2, 2, 2, 2>), dim3((unsigned)ntiles), dim3(256), 0, st, d_tiles, d_groups, d_prods, ntiles);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

}  // namespace dmrgx
