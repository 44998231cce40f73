// Grouped f64 GEMM for gfx950 (MI355X): 64x64 output tile per 256-thread workgroup, 4 waves in a 2x2
// arrangement, each wave owning a 32x32 sub-tile as 2x2 v_mfma_f64_16x16x4_f64 accumulators.
//
// Operand staging (HBM -> registers -> LDS, one barrier per 16-deep k-step, register prefetch of the next
// k-step while the current one is multiplied):
//   A tile 64 x 16, stored [i][k] with row stride 18 doubles  -> the MFMA A fragment (lane: i=l&15, k=l>>4)
//                                                                 reads all 64 LDS banks exactly once per
//                                                                 32-lane ds_read_b64 pass
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

// LDS strides are derived inside the kernel template: A rows 18 doubles apart (the fragments are read with explicit
// ds_read_b64: within a 32-lane pass the addresses i*18 + {k, k+1}, i < 16, cover every 8-byte bank pair once; an odd
// stride always collides for one (i, i') pair, measured 20 % conflict cycles at 17), B rows BN+16 doubles apart.

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
    constexpr int AS_LD = BK + 2;        // 18: the 32 lanes of a ds_read_b64 pass (i < 16, k and k+1) hit 32 distinct bank pairs
    constexpr int BS_LD = BN + 16;       // == 32 dwords (mod 64): rows k and k+1 use disjoint bank halves
    constexpr int AROWS = THREADS / 16;  // A rows covered per pass
    constexpr int NA = BM / AROWS;       // A elements per thread per k-step
    constexpr int BROWS = THREADS / BN;  // B rows covered per pass
    constexpr int NB = BK / BROWS;       // B elements per thread per k-step
    __shared__ double As[2][BM * AS_LD];
    __shared__ double Bs[2][BK * BS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    const GTile tile = tiles[blockIdx.x];
    if (tile.group < 0) return;
    const GGroup g = groups[tile.group];
    const int m0 = tile.tm * GG_BM, n0 = tile.tn * GG_BN;        // tile coordinates are in 64-units for both shapes
    const int mrem = min(BM, g.M - m0), nrem = min(BN, g.N - n0);
    const int wr = wave / WC, wc = wave % WC;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int wrow = wr * 16 * TR, wcol = wc * 16 * TC;           // wave sub-tile origin
    // 16 x 16 accumulator blocks of this wave that intersect the output (edge tiles): blocks outside are never
    // multiplied, so ragged sector sizes cost MFMA time at 16-granularity, not at tile granularity (wave-uniform).
    const int tr_eff = min(TR, max(0, (mrem - wrow + 15) >> 4)), tc_eff = min(TC, max(0, (nrem - wcol + 15) >> 4));
    const bool wave_full = (tr_eff == TR) && (tc_eff == TC);

    d4 acc[TR][TC];
#pragma unroll
    for (int mi = 0; mi < TR; ++mi)
#pragma unroll
        for (int ni = 0; ni < TC; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

    // ---- GEMM stream ------------------------------------------------------------------------------------
    // The MFMA pipe and the ordinary VALU do not co-issue on a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0 on gfx950), so every
    // vector integer instruction in the k-step loop is MFMA time lost.  The steady-state k-step therefore contains no
    // VALU work besides the MFMAs: per-thread byte offsets are computed once per product (aoff/boff), the wave-uniform
    // base advances in SGPRs, and the LDS buffer index is a compile-time constant (loop unrolled by two) so that all LDS
    // addresses are loop invariants.
    const int a_r = tid >> 4, a_k = tid & 15;      // A loader: rows a_r + AROWS s, column a_k (128 B per 16 lanes)
    const int b_k = tid / BN, b_j = tid % BN;      // B loader: rows b_k + BROWS s, column b_j (512 B per wave)
    // Ragged M / N edges need no masking: a row of A beyond mrem only feeds rows of C beyond mrem, a column of B beyond
    // nrem only columns beyond nrem, and the epilogue never stores those -- the loaders just clamp to the last valid
    // row / column so that every address is in bounds.  Only the K edge is zeroed (last k-step of a product).
    const int bcol = min(b_j, nrem - 1);
    unsigned aoff[NA], boff[NB];                   // per-thread byte offsets inside the current product's panels
    double ra[NA], rb[NB];
    const int pend = g.prod_end;
    int p = g.prod_begin + g.n_axpy;
    int k0 = 0;
    const double *cA = nullptr, *cB = nullptr;     // current product, wave-uniform -> SGPRs
    int clda = 0, cldb = 0, cK = 0;

// Operand pointers come out of the task table, so the compiler would treat them as generic and emit flat_load
// (+ lgkmcnt waits that serialise against LDS); they are global by construction -> explicit address space, and the
// loads are "uniform base + 32-bit per-thread offset".
#define GG_PRODUCT(P)                                                                         \
    {                                                                                         \
        cA = prods[P].A; cB = prods[P].B; clda = prods[P].lda; cldb = prods[P].ldb; cK = prods[P].K; \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) aoff[s] = ((unsigned)min(a_r + AROWS * s, mrem - 1) * (unsigned)clda + (unsigned)a_k) * 8u; \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) boff[s] = ((unsigned)(b_k + BROWS * s) * (unsigned)cldb + (unsigned)bcol) * 8u; \
    }
#define GG_GLOAD(KK0)                                                                         \
    {                                                                                         \
        gbptr A_ = (gbptr)(cA + (size_t)m0 * clda + (KK0));                                   \
        gbptr B_ = (gbptr)(cB + (size_t)(KK0) * cldb + n0);                                   \
        const int klast_ = cK - 1 - (KK0);                                                    \
        if (klast_ >= BK - 1) {                                                               \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) ra[s] = *(gptr)(A_ + aoff[s]);     \
            _Pragma("unroll") for (int s = 0; s < NB; ++s) rb[s] = *(gptr)(B_ + boff[s]);     \
        } else {                                                                              \
            const double ak_ = a_k <= klast_ ? 1.0 : 0.0;                                     \
            const unsigned aback_ = (unsigned)max(a_k - klast_, 0) * 8u;                      \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) ra[s] = *(gptr)(A_ + (aoff[s] - aback_)) * ak_; \
            _Pragma("unroll") for (int s = 0; s < NB; ++s) {                                  \
                const int k_ = b_k + BROWS * s;                                               \
                rb[s] = *(gptr)(B_ + ((unsigned)min(k_, klast_) * (unsigned)cldb + (unsigned)bcol) * 8u) * (k_ <= klast_ ? 1.0 : 0.0); \
            }                                                                                 \
        }                                                                                     \
    }
#define GG_LSTORE(BUF)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) As[BUF][(a_r + AROWS * s) * AS_LD + a_k] = ra[s]; \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) Bs[BUF][(b_k + BROWS * s) * BS_LD + b_j] = rb[s]; \
    }

// MFMA fragments come from LDS through explicit ds_read_b64 with immediate offsets off two per-tile base registers
// (left to itself the compiler rematerialises a v_add_u32 per fragment address inside the loop).  GG_FRAG_WAIT is the
// matching s_waitcnt, tied to the fragment registers so that the MFMAs cannot be scheduled above it.
    const unsigned lds_a = (unsigned)(size_t)&As[0][(wrow + l15) * AS_LD + l4];
    const unsigned lds_b = (unsigned)(size_t)&Bs[0][l4 * BS_LD + wcol + l15];
    double fa[2][TR], fb[2][TC];
#define GG_FRAG(SET, BUF, KK)                                                                 \
    {                                                                                         \
        _Pragma("unroll") for (int mi = 0; mi < TR; ++mi)                                     \
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[SET][mi]) : "v"(lds_a), "i"(8 * ((BUF) * BM * AS_LD + 16 * AS_LD * mi + (KK))) : "memory"); \
        _Pragma("unroll") for (int ni = 0; ni < TC; ++ni)                                     \
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[SET][ni]) : "v"(lds_b), "i"(8 * ((BUF) * BK * BS_LD + (KK) * BS_LD + 16 * ni)) : "memory"); \
    }
#define GG_FRAG_WAIT(SET)                                                                     \
    {                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
        _Pragma("unroll") for (int mi = 0; mi < TR; ++mi) asm volatile("" : "+v"(fa[SET][mi])); \
        _Pragma("unroll") for (int ni = 0; ni < TC; ++ni) asm volatile("" : "+v"(fb[SET][ni])); \
    }
#define GG_MFMA(SET, GUARD, I0, I1)      /* MFMAs number I0 .. I1-1 (row-major over the TR x TC blocks) of one k-group */ \
    {                                                                                         \
        _Pragma("unroll") for (int q = (I0); q < (I1); ++q) {                                 \
            const int mi = q / TC, ni = q % TC;                                               \
            if (GUARD) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][mi], fb[SET][ni], acc[mi][ni], 0, 0, 0); \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }

    bool have = p < pend;
    if (have) {
        GG_PRODUCT(p);
        GG_GLOAD(0);
        GG_LSTORE(0);
    }
    __syncthreads();
    if (have) GG_FRAG(0, 0, 0);
// One k-step (4 k-groups of 4) on LDS buffer BUF.  The waves of a SIMD advance in lockstep (the MFMA arbiter is fair:
// with n waves each gets the pipe every n-th MFMA), so a wave has about (n-1) x 64 cycles of slack after each of its
// MFMAs, and whatever it does between two MFMAs beyond that slack is MFMA time lost on the whole SIMD.  The step is
// therefore laid out by hand, with the non-MFMA work cut into pieces that sit between individual MFMAs:
//   k-group 0: bookkeeping for the next k-step | its global loads (-> registers) | fragment reads of group 1
//   k-group 1: fragment reads of group 2
//   k-group 2: park the prefetched registers in buffer BUF^1 | fragment reads of group 3 | barrier | first fragments of
//              the next k-step (from BUF^1)
//   k-group 3: nothing but MFMAs
// Every fragment read of buffer BUF is issued and waited for before the barrier, so a fast wave that goes on to
// refill BUF one step later cannot overtake a reader.  sched_barrier pins the order against the compiler's scheduler.
#define GG_STEP(BUF, GUARD)                                                                   \
    {                                                                                         \
        constexpr int NM = TR * TC;                                                           \
        GG_FRAG_WAIT(0);                                                                      \
        GG_MFMA(0, GUARD, 0, 1);                                                              \
        int pn = p, kn = k0 + BK;                                                             \
        if (kn >= cK) { pn = p + 1; kn = 0; }                                                 \
        const bool have_next = pn < pend;                                                     \
        if (have_next && pn != p) GG_PRODUCT(pn);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, 2);                                                              \
        if (have_next) GG_GLOAD(kn);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 2, NM - 1);                                                         \
        GG_FRAG(1, BUF, 4);                                                                   \
        GG_MFMA(0, GUARD, NM - 1, NM);                                                        \
        GG_FRAG_WAIT(1);                                                                      \
        GG_MFMA(1, GUARD, 0, NM - 1);                                                         \
        GG_FRAG(0, BUF, 8);                                                                   \
        GG_MFMA(1, GUARD, NM - 1, NM);                                                        \
        GG_FRAG_WAIT(0);                                                                      \
        GG_MFMA(0, GUARD, 0, 1);                                                              \
        if (have_next) GG_LSTORE((BUF) ^ 1);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, NM - 1);                                                         \
        GG_FRAG(1, BUF, 12);                                                                  \
        GG_MFMA(0, GUARD, NM - 1, NM);                                                        \
        GG_FRAG_WAIT(1);                                                                      \
        __syncthreads();                                                                      \
        if (have_next) GG_FRAG(0, (BUF) ^ 1, 0);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(1, GUARD, 0, NM);                                                             \
        p = pn; k0 = kn; have = have_next;                                                    \
    }
// The k-step stream, instantiated twice: interior waves run it with every MFMA unconditional; waves on a ragged tile
// edge run a copy whose MFMAs are guarded by wave-uniform block bounds (blocks outside the output are never multiplied).
// Both copies execute the same barriers, so the waves of one workgroup may take different copies.  (One loop with a
// per-k-step choice makes the compiler merge the accumulators with v_mov copies that wait on the MFMA results.)
#define GG_STREAM(GUARD)                                                                      \
    while (have) {                                                                            \
        GG_STEP(0, GUARD);                                                                    \
        if (!have) break;                                                                     \
        GG_STEP(1, GUARD);                                                                    \
    }
    if (TR * TC > 6 || wave_full) { GG_STREAM(true) }     // (the 128 x 128 kernel only ever gets interior tiles; a 96-row tile is ragged by construction)
    else { GG_STREAM(mi < tr_eff && ni < tc_eff) }
#undef GG_STREAM
#undef GG_STEP
#undef GG_FRAG
#undef GG_FRAG_WAIT
#undef GG_MFMA
#undef GG_PRODUCT
#undef GG_GLOAD
#undef GG_LSTORE

    // ---- scaled-copy products (identity operator cells): acc += alpha * S[tile] ------------------------
    // (after the GEMM stream: its 64 value registers then do not meet the stream's operand / fragment registers)
    // Two products per pass with all their loads issued before the first use: a scaled copy is a descriptor load followed by
    // a tile load, and one at a time the chain of both latencies (~2.5 us) is paid per product.
    int q = g.prod_begin;
    const int axpy_end = q + g.n_axpy;
    if (TR * TC <= 4) {
        for (; q + 1 < axpy_end; q += 2) {
            const GProd pa = prods[q], pb = prods[q + 1];
            gptr Sa = (gptr)(pa.B + (size_t)m0 * pa.ldb + n0);
            gptr Sb = (gptr)(pb.B + (size_t)m0 * pb.ldb + n0);
            double va[TR][TC][4], vb[TR][TC][4];
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = min(wrow + mi * 16 + l4 + 4 * r, mrem - 1), col = min(wcol + ni * 16 + l15, nrem - 1);   // clamped: in bounds, never stored
                        va[mi][ni][r] = Sa[(size_t)row * pa.ldb + col];
                        vb[mi][ni][r] = Sb[(size_t)row * pb.ldb + col];
                    }
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc[mi][ni][r] += pa.alpha * va[mi][ni][r]; acc[mi][ni][r] += pb.alpha * vb[mi][ni][r]; }
        }
    }
    for (; q < axpy_end; ++q) {
        const GProd pr = prods[q];
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

int ggemm_cluster()
{
    static const int c = [] { const char* e = getenv("DMRGX_CLUSTER"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 16 ? v : 8; }();
    return c;
}

bool ggemm_use_big_tiles()
{
    const char* e = getenv("DMRGX_TILES");
    return e && (std::string(e) == "mixed" || std::string(e) == "overlap");   // overlap: mixed tiles, the two launches of a stage on two streams
}

void ggemm_schedule(std::vector<GTile>& tiles, int unit)
{
    constexpr int NX = 8;
    if (tiles.empty()) return;
    struct Cl { size_t begin, end; int64_t cost; };
    std::vector<Cl> cl;
    for (size_t i = 0; i < tiles.size();) {
        size_t j = i;
        int64_t c = 0;
        // a cluster = the consecutive tiles of one group inside one GG_CLUSTER x GG_CLUSTER block of its tile grid
        const int W = GG_CLUSTER * unit, cm = tiles[i].tm / W, cn = tiles[i].tn / W;
        while (j < tiles.size() && tiles[j].group == tiles[i].group && tiles[j].tm / W == cm && tiles[j].tn / W == cn && j - i < 256) { c += tiles[j].pad + 2; ++j; }
        cl.push_back(Cl{i, j, c});
        i = j;
    }
    std::stable_sort(cl.begin(), cl.end(), [](const Cl& a, const Cl& b) { return a.cost > b.cost; });
    // clusters go to the least-loaded XCD, heaviest first (LPT) ...
    std::vector<std::vector<Cl>> binc(NX);
    int64_t load[NX] = {0};
    for (const Cl& c : cl) {
        int best = 0;
        for (int x = 1; x < NX; ++x) if (load[x] < load[best]) best = x;
        load[best] += c.cost;
        binc[best].push_back(c);
    }
    // ... and inside an XCD the clusters with the LONGEST tiles run first, whatever their total: the launch ends when the
    // last tile ends, so the tail should be made of the shortest tiles, not of a small cluster of long ones
    static const bool by_tile = !(getenv("DMRGX_SCHED_TOTAL"));
    std::vector<std::vector<GTile>> bins(NX);
    for (int x = 0; x < NX; ++x) {
        // (coarse buckets of 16 k-steps, stable: clusters of similar tile length keep the LPT order, which keeps the clusters
        //  of one group -- same operands -- close together)
        if (by_tile) std::stable_sort(binc[x].begin(), binc[x].end(), [&](const Cl& a, const Cl& b) { return (tiles[a.begin].pad >> 4) > (tiles[b.begin].pad >> 4); });
        for (const Cl& c : binc[x]) for (size_t t = c.begin; t < c.end; ++t) bins[x].push_back(tiles[t]);
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
    if (big == GG_SHAPE_TALL) hipLaunchKernelGGL((ggemm_kernel<3, 2, 2, 2>), dim3((unsigned)ntiles), dim3(256), 0, st, d_tiles, d_groups, d_prods, ntiles);
    else if (big) hipLaunchKernelGGL((ggemm_kernel<4, 2, 2, 4>), dim3((unsigned)ntiles), dim3(512), 0, st, d_tiles, d_groups, d_prods, ntiles);
    else hipLaunchKernelGGL((ggemm_kernel<2, 2, 2, 2>), dim3((unsigned)ntiles), dim3(256), 0, st, d_tiles, d_groups, d_prods, ntiles);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

}  // namespace dmrgx
