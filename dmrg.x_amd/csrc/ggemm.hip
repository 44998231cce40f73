// Grouped f64 GEMM for gfx950 (MI355X): 64x64 output tile per 256-thread workgroup, 4 waves in a 2x2
// arrangement, each wave owning a 32x32 sub-tile as 2x2 v_mfma_f64_16x16x4_f64 accumulators.
//
// Operand staging (HBM -> registers -> LDS, one barrier per 16-deep k-step, register prefetch of the next
// k-step while the current one is multiplied):
//   A tile 64 x 16, stored [i][k] with row stride 18 doubles  -> the MFMA A fragment (lane: i=l&15, k=l>>4)
//                                                                 reads all 64 LDS banks exactly once
//   B tile 16 x 64, stored [k][j] with row stride 80 doubles  -> same for the B fragment (k=l>>4, j=l&15)
// C/D fragment of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg  (verified on hardware,
// profiles/r01_mfma_f64_probe.txt).
//
// The k-steps of all GEMM products of a group form one software-pipelined stream, so a group made of many
// short products (the superblock MatMult's stage 2: up to ~26 operator cells per output tile) keeps the
// pipeline full across product boundaries.
#include "ggemm.h"

namespace dmrgx {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int AS_LD = GG_BK + 2;    // 18: (2*AS_LD/4) odd -> 16 rows hit 16 distinct 4-bank groups
constexpr int BS_LD = GG_BN + 16;   // 80: 160 dwords == 32 (mod 64) -> k and k+1 rows use disjoint bank halves

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous chunk of the tile list so that
// tiles sharing operator cells / wavefunction panels hit the same 4 MiB L2 (bijective for any n).
__device__ __forceinline__ int xcd_chunk_index(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

__global__ void __launch_bounds__(GG_THREADS)
ggemm_kernel(const GTile* __restrict__ tiles, const GGroup* __restrict__ groups, const GProd* __restrict__ prods, int ntiles)
{
    __shared__ double As[2][GG_BM * AS_LD];
    __shared__ double Bs[2][GG_BK * BS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GTile tile = tiles[xcd_chunk_index(blockIdx.x, ntiles)];
    const GGroup g = groups[tile.group];
    const int m0 = tile.tm * GG_BM, n0 = tile.tn * GG_BN;
    const int mrem = min(GG_BM, g.M - m0), nrem = min(GG_BN, g.N - n0);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    d4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

    // ---- scaled-copy products (identity operator cells): acc += alpha * S[tile] ------------------------
    int p = g.prod_begin;
    for (int e = p + g.n_axpy; p < e; ++p) {
        const GProd pr = prods[p];
        const double* S = pr.B + (size_t)m0 * pr.ldb + n0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr * 32 + mi * 16 + l4 + 4 * r, col = wc * 32 + ni * 16 + l15;
                    if (row < mrem && col < nrem) acc[mi][ni][r] += pr.alpha * S[(size_t)row * pr.ldb + col];
                }
    }

    // ---- GEMM stream ------------------------------------------------------------------------------------
    const int a_r = tid >> 4, a_k = tid & 15;   // A loader: rows a_r + 16 s, column a_k   (128 B per 16 lanes)
    const int b_k = tid >> 6, b_j = tid & 63;   // B loader: rows b_k + 4 s,  column b_j   (512 B per wave)
    double ra[4], rb[4];
    const int pend = g.prod_end;
    int k0 = 0;
    // current product, kept as scalars (wave-uniform -> SGPRs)
    const double *cA = nullptr, *cB = nullptr;
    int clda = 0, cldb = 0, cK = 0;

#define GG_GLOAD(PA, PB, LDA, LDB, KK, KK0)                                                   \
    {                                                                                         \
        const double* A_ = (PA) + (size_t)m0 * (LDA) + (KK0);                                 \
        const double* B_ = (PB) + (size_t)(KK0) * (LDB) + n0;                                 \
        const bool ak_ = ((KK0) + a_k) < (KK), bj_ = b_j < nrem;                              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                       \
            const int row_ = a_r + 16 * s;                                                    \
            ra[s] = (ak_ && row_ < mrem) ? A_[(size_t)row_ * (LDA) + a_k] : 0.0;              \
            const int k_ = b_k + 4 * s;                                                       \
            rb[s] = (bj_ && ((KK0) + k_) < (KK)) ? B_[(size_t)k_ * (LDB) + b_j] : 0.0;        \
        }                                                                                     \
    }
#define GG_LSTORE(BUF)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                       \
            As[BUF][(a_r + 16 * s) * AS_LD + a_k] = ra[s];                                    \
            Bs[BUF][(b_k + 4 * s) * BS_LD + b_j] = rb[s];                                     \
        }                                                                                     \
    }

    bool have = p < pend;
    if (have) {
        cA = prods[p].A; cB = prods[p].B; clda = prods[p].lda; cldb = prods[p].ldb; cK = prods[p].K;
        GG_GLOAD(cA, cB, clda, cldb, cK, 0);
        GG_LSTORE(0);
    }
    __syncthreads();
    int buf = 0;
    while (have) {
        int pn = p, kn = k0 + GG_BK;
        if (kn >= cK) { pn = p + 1; kn = 0; }
        const bool have_next = pn < pend;
        if (have_next) {
            if (pn != p) { cA = prods[pn].A; cB = prods[pn].B; clda = prods[pn].lda; cldb = prods[pn].ldb; cK = prods[pn].K; }
            GG_GLOAD(cA, cB, clda, cldb, cK, kn);
        }
        const double* as = &As[buf][(wr * 32 + l15) * AS_LD + l4];
        const double* bs = &Bs[buf][l4 * BS_LD + wc * 32 + l15];
#pragma unroll
        for (int kk = 0; kk < GG_BK; kk += 4) {
            const double a0 = as[kk], a1 = as[16 * AS_LD + kk];
            const double b0 = bs[kk * BS_LD], b1 = bs[kk * BS_LD + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (have_next) GG_LSTORE(buf ^ 1);
        __syncthreads();
        buf ^= 1; p = pn; k0 = kn; have = have_next;
    }
#undef GG_GLOAD
#undef GG_LSTORE

    // ---- epilogue ---------------------------------------------------------------------------------------
    double* C = g.C + (size_t)m0 * g.ldc + n0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 32 + mi * 16 + l4 + 4 * r, col = wc * 32 + ni * 16 + l15;
                if (row < mrem && col < nrem) {
                    double* c = C + (size_t)row * g.ldc + col;
                    *c = g.accumulate ? (*c + acc[mi][ni][r]) : acc[mi][ni][r];
                }
            }
}

dmrgx_status ggemm_launch(const GTile* d_tiles, const GGroup* d_groups, const GProd* d_prods, int32_t ntiles, hipStream_t st)
{
    if (ntiles <= 0) return DMRGX_OK;
    hipLaunchKernelGGL(ggemm_kernel, dim3((unsigned)ntiles), dim3(GG_THREADS), 0, st, d_tiles, d_groups, d_prods, ntiles);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

}  // namespace dmrgx
