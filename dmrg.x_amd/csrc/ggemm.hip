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
#include <atomic>
#include <cstdlib>
#include <mutex>

namespace dmrgx {

typedef double d4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DDMRGX_TILE_TRACE, tools/tile_trace.sh; never compiled into the product library): every workgroup stamps
// the 100 MHz real-time counter at the milestones of its tile into a buffer of its own -- no output value depends on a stamp.
#ifdef DMRGX_TILE_TRACE
#define GG_TRACE_PARAM , unsigned long long* __restrict__ trace
#define GG_STAMP(I) { if (trace && threadIdx.x == 0) trace[(size_t)t * 8 + (I)] = __builtin_amdgcn_s_memrealtime(); }
#define GG_STAMP_ID() { if (trace && threadIdx.x == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); trace[(size_t)t * 8 + 7] = ((unsigned long long)xcc << 32) | hw; } }
#define GG_CLK0() unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
#define GG_CLK1() { if (trace && threadIdx.x == 0) trace[(size_t)t * 8 + 6] = __builtin_amdgcn_s_memtime() - clk0_; }   /* shader cycles of the k-step stream */
#else
#define GG_TRACE_PARAM
#define GG_STAMP(I)
#define GG_STAMP_ID()
#define GG_CLK0()
#define GG_CLK1()
#endif
typedef const double __attribute__((address_space(1)))* gptr;   // global-memory pointer (see GG_GLOAD)
typedef double __attribute__((address_space(1)))* gwptr;
typedef const char __attribute__((address_space(1)))* gbptr;    // byte pointer for base + 32-bit offset addressing

// The task tables are written before the launch and never during it: they are read through the CONSTANT address space, which is
// what makes their (wave-uniform) loads scalar loads into SGPRs wherever they stand.  Through a generic pointer the compiler may
// only do that while no store of the kernel can precede the load; in a resident workgroup the descriptor loads of tile t + 1 follow
// the output stores of tile t, they became vector loads, and every branch on a descriptor field a divergent one (measured on the
// first build of this kernel: 164 exec-mask branches and 405 spilled registers instead of 52 and 12).
typedef const GTile __attribute__((address_space(4)))* ktile_ptr;
typedef const GGroup __attribute__((address_space(4)))* kgroup_ptr;
typedef const GProd __attribute__((address_space(4)))* kprod_ptr;
__device__ __forceinline__ GTile kload(ktile_ptr p, int i) { GTile r; r.group = p[i].group; r.tm = p[i].tm; r.tn = p[i].tn; r.pad = p[i].pad; return r; }
__device__ __forceinline__ GGroup kload(kgroup_ptr p, int i)
{
    GGroup r;
    r.C = p[i].C; r.ldc = p[i].ldc; r.M = p[i].M; r.N = p[i].N; r.prod_begin = p[i].prod_begin; r.prod_end = p[i].prod_end;
    r.n_axpy = p[i].n_axpy; r.accumulate = p[i].accumulate;
    return r;
}
__device__ __forceinline__ GProd kload(kprod_ptr p, int i)
{
    GProd r;
    r.A = p[i].A; r.B = p[i].B; r.lda = p[i].lda; r.ldb = p[i].ldb; r.K = p[i].K; r.kind = p[i].kind; r.alpha = p[i].alpha;
    return r;
}

// LDS strides are derived inside the kernel template: A rows 18 doubles apart (the fragments are read with explicit
// ds_read_b64: within a 32-lane pass the addresses i*18 + {k, k+1}, i < 16, cover every 8-byte bank pair once; an odd
// stride always collides for one (i, i') pair, measured 20 % conflict cycles at 17), B rows BN+16 doubles apart.

// Workgroup -> tiles: a launch has G = min(ntiles, slots of the chip) RESIDENT workgroups; workgroup w walks the list entries
// w, w + G, w + 2G, ... (its own tile list, laid out by the host: ggemm_schedule) until the first entry with group < 0.  Blocks
// b, b+8, b+16, ... -- the workgroups the dispatcher deals to one XCD -- own that XCD's cost-balanced, locality-clustered lists.
//
// Why resident workgroups (round 4; tools/tile_trace.sh on cfg4real, profiles/r04_tile_trace_before.txt): with one workgroup per
// tile a stage-1 launch kept 3.60 of 4 workgroup slots per CU occupied (7.5 % of the slot time lay between the end of one
// workgroup and the start of the next on the same CU: same-length tiles of a cluster end together and the dispatcher refills the
// slots one after the other, median 1.8 us, p90 16.6 us) and a resident workgroup spent 13.6 % of its time outside the k-step
// stream (descriptor chain 1.2 us, first operands 1.6 us, scaled copies 2.4 us, stores 3.0 us per 52 us tile), so only 3.12
// workgroups per CU were feeding the MFMA pipe; the k-step itself runs at the pipe's limit (1.95 us for a full tile with four
// streams on the CU).  A resident workgroup (a) is never re-dispatched, (b) has the next tile's descriptors in SGPRs a whole
// tile ahead, and (c) issues the next tile's first operand loads in the LAST k-step of the current tile, so that the switch
// costs the epilogue only.

// TR x TC v_mfma_f64_16x16x4 accumulators per wave, WR x WC waves per workgroup: tile (16 TR WR) x (16 TC WC).
//   <2,2,2,2>:  64 x  64 tile, 256 threads, 4 workgroups per CU -- ragged remainders and small sectors
//   <4,2,2,4>: 128 x 128 tile, 512 threads, 2 workgroups per CU -- same 4 waves per SIMD, half the L2->LDS bytes
//                                                                 per flop, 64 accumulator registers per wave
template <int TR, int TC, int WR, int WC>
__global__ void __launch_bounds__(64 * WR * WC, 4)
ggemm_kernel(const GTile* __restrict__ tiles_, const GGroup* __restrict__ groups_, const GProd* __restrict__ prods_, int ntiles, int* __restrict__ ctr GG_TRACE_PARAM)
{
    const ktile_ptr tiles = (ktile_ptr)tiles_;
    const kgroup_ptr groups = (kgroup_ptr)groups_;
    const kprod_ptr prods = (kprod_ptr)prods_;
    constexpr int THREADS = 64 * WR * WC;
    constexpr int BM = 16 * TR * WR, BN = 16 * TC * WC, BK = GG_BK;
    constexpr int AS_LD = BK + 2;        // 18: the 32 lanes of a ds_read_b64 pass (i < 16, k and k+1) hit 32 distinct bank pairs
    constexpr int BS_LD = BN + 16;       // == 32 dwords (mod 64): rows k and k+1 use disjoint bank halves
    constexpr int AROWS = THREADS / 16;  // A rows covered per pass
    constexpr int NA = BM / AROWS;       // A elements per thread per k-step
    constexpr int BROWS = THREADS / BN;  // B rows covered per pass
    constexpr int NB = BK / BROWS;       // B elements per thread per k-step
    constexpr bool RELAYOUT = (TR == 2 && TC == 2 && WR == 2 && WC == 2);   // thin edge tiles: blocks dealt over all four waves
    __shared__ double As[2][BM * AS_LD];
    __shared__ double Bs[2][BK * BS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    // Work distribution: the list is eight interleaved per-XCD queues (entry i belongs to XCD i & 7; the host balances the queues).
    // Workgroup b starts on entry b and then claims the further entries of its XCD's queue, in order, with an atomic counter
    // (ctr[0..7], all zero at launch and reset by the last workgroup to leave; ctr[8] counts the leavers).  The claim for the NEXT
    // tile is issued at the start of the current one and read after its k-step stream, so its latency is never waited for.
    __shared__ int sh_next;
    const int G = gridDim.x;
    const bool dyn = ntiles > G;                   // (else every entry has its own workgroup)
    const int xcd = blockIdx.x & 7, qbase = G >> 3;
    int t = blockIdx.x;
    GTile tl = kload(tiles, t);
    if (tl.group < 0) {                             // padding: this queue is shorter than its share of starting tiles
        if (dyn && tid == 0 && atomicAdd(&ctr[8], 1) == G - 1) { for (int i = 0; i < 9; ++i) ctr[i] = 0; }
        return;
    }
    GGroup g = kload(groups, tl.group);
    const GTile tl_none = GTile{-1, 0, 0, 0};
#ifdef DMRGX_STAGGER
    if (dyn) for (int i = ((blockIdx.x >> 3) & 63) * DMRGX_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(8);   // experiment: ~250 ns per step
#endif
    // Per-lane constants.  Only the two LDS store bases (and, per tile, the two fragment bases) are kept in registers across the
    // k-step stream; everything else that depends on the lane is recomputed where it is used from an OPAQUE copy of the thread
    // index (GG_LANES): left visible, the compiler computed some twenty such values once, ran out of registers and kept them in
    // scratch, reloading them around every tile.
    //   A loader: rows a_r + AROWS s, column a_k (128 B per 16 lanes);  B loader: rows b_k + BROWS s, column b_j (512 B per wave)
#define GG_LANES()                                                                            \
    int tp_ = tid;                                                                            \
    asm volatile("" : "+v"(tp_));                                                             \
    const int a_r = tp_ >> 4, a_k = tp_ & 15, b_k = tp_ / BN, b_j = tp_ % BN;                 \
    (void)a_r; (void)a_k; (void)b_k; (void)b_j;
    unsigned st_a, st_b;                           // LDS byte addresses of this lane's first A / B staging element (buffer 0)
    {
        GG_LANES();
        st_a = (unsigned)(size_t)&As[0][a_r * AS_LD + a_k];
        st_b = (unsigned)(size_t)&Bs[0][b_k * BS_LD + b_j];
    }

    // Loader state: the tile whose operands are being fetched.  It runs one k-step ahead of the multiplication and therefore
    // moves on to the NEXT tile during the last k-step of the current one.
    int lm0, ln0, lmrem, lnrem;
    unsigned aoff[NA], boff[NB];                   // per-thread byte offsets inside the current product's panels
    double ra[NA], rb[NB];
    const double *cA = nullptr, *cB = nullptr;     // current product, wave-uniform -> SGPRs
    int clda = 0, cldb = 0, cK = 0;
    int kz = 0;                                    // K edge of the k-step in the registers: > 0 columns below kz, < 0 columns from -kz on are zeroed
    bool pre = false;                              // first k-step of tile `tl` already fetched (registers -> LDS buffer 0 below)

// Operand pointers come out of the task table, so the compiler would treat them as generic and emit flat_load
// (+ lgkmcnt waits that serialise against LDS); they are global by construction -> explicit address space, and the
// loads are "uniform base + 32-bit per-thread offset".
// Ragged M / N edges need no masking: a row of A beyond mrem only feeds rows of C beyond mrem, a column of B beyond
// nrem only columns beyond nrem, and the epilogue never stores those -- the loaders just clamp to the last valid
// row / column so that every address is in bounds.  Only the K edge is zeroed (last k-step of a product).
#define GG_LOADER_TILE(TL, GR)                                                                \
    {                                                                                         \
        lm0 = (TL).tm * GG_BM; ln0 = (TL).tn * GG_BN;                                         \
        lmrem = min(BM, (GR).M - lm0);                                                        \
        lnrem = min(BN, (GR).N - ln0);                                                        \
    }
// K edge (last k-step of a product, K not a multiple of 16) without a second load path: the wave-uniform base is pulled back so that
// the 16 columns fetched are the LAST 16 of the product -- all in bounds -- and the columns that were multiplied already (the first
// `kz` of them) are zeroed in the A registers on their way into LDS (A zero is enough: the B rows beside them are real, finite
// operand rows).  A product shorter than one k-step clamps its per-thread offsets to its last column / row instead (computed once
// per product) and zeroes the clamped duplicates (columns above `kz`).  Either way the k-step issues the same eight loads off
// loop-invariant offset registers; the zeroing is eight selects under a wave-uniform branch, in the edge step only.
#define GG_PRODUCT_SET(PR)                                                                    \
    {                                                                                         \
        cA = (PR).A; cB = (PR).B; clda = (PR).lda; cldb = (PR).ldb; cK = (PR).K;              \
        const int kc_ = min(cK, BK) - 1;                                                      \
        GG_LANES();                                                                           \
        const int bcol = min(b_j, lnrem - 1);                                                 \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) aoff[s] = ((unsigned)min(a_r + AROWS * s, lmrem - 1) * (unsigned)clda + (unsigned)min(a_k, kc_)) * 8u; \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) boff[s] = ((unsigned)min(b_k + BROWS * s, kc_) * (unsigned)cldb + (unsigned)bcol) * 8u; \
    }
#define GG_PRODUCT(P) { const GProd pr_ = kload(prods, P); GG_PRODUCT_SET(pr_); }
#define GG_GLOAD(KK0)                                                                         \
    {                                                                                         \
        const int klast_ = cK - 1 - (KK0);                                                    \
        const int back_ = (klast_ < BK - 1 && cK >= BK) ? BK - 1 - klast_ : 0;                \
        kz = klast_ < BK - 1 ? (cK >= BK ? back_ : -(klast_ + 1)) : 0;                        \
        gbptr A_ = (gbptr)(cA + (size_t)lm0 * clda + ((KK0) - back_));                        \
        gbptr B_ = (gbptr)(cB + (size_t)((KK0) - back_) * cldb + ln0);                        \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) ra[s] = *(gptr)(A_ + aoff[s]);         \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) rb[s] = *(gptr)(B_ + boff[s]);         \
    }
#define GG_LSTORE(BUF)                                                                        \
    {                                                                                         \
        if (kz != 0) {                                                                        \
            GG_LANES();                                                                       \
            const bool z_ = kz > 0 ? a_k < kz : a_k >= -kz;                                   \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) ra[s] = z_ ? 0.0 : ra[s];          \
        }                                                                                     \
        _Pragma("unroll") for (int s = 0; s < NA; ++s)                                        \
            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(st_a), "v"(ra[s]), "i"(8 * ((BUF) * BM * AS_LD + AROWS * s * AS_LD)) : "memory"); \
        _Pragma("unroll") for (int s = 0; s < NB; ++s)                                        \
            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(st_b), "v"(rb[s]), "i"(8 * ((BUF) * BK * BS_LD + BROWS * s * BS_LD)) : "memory"); \
    }

// MFMA fragments come from LDS through explicit ds_read_b64 with immediate offsets off two per-tile base registers
// (left to itself the compiler rematerialises a v_add_u32 per fragment address inside the loop).  GG_FRAG_WAIT is the
// matching s_waitcnt, tied to the fragment registers so that the MFMAs cannot be scheduled above it.
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
            GUARD(acc[mi][ni], fa[SET][mi], fb[SET][ni], q);                                  \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
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

    for (;;) {
        GG_STAMP(0);
        // ---- the tile being multiplied --------------------------------------------------------------------------
        const int m0 = tl.tm * GG_BM, n0 = tl.tn * GG_BN;            // tile coordinates are in 64-units for both shapes
        const int mrem = min(BM, g.M - m0), nrem = min(BN, g.N - n0);
        // wave -> 16 x 16 accumulator blocks.  Default: WR x WC waves, each TR x TC blocks.  A thin edge tile (at most 32 valid
        // rows or columns) would leave half of the waves without work and give the other half two blocks each -- a chain of
        // MFMAs twice as long, each of which queues behind the MFMAs of the three full tiles on the same CU (measured: a
        // sliver held its slot 70 % as long as a full tile for <= 25 % of the work); its blocks are dealt over all four waves.
        int wrow = (wave / WC) * 16 * TR, wcol = (wave % WC) * 16 * TC;
        int tr_eff, tc_eff;
        if (RELAYOUT && mrem <= 32) {            // 1 x 4 waves: both block rows, block column `wave`
            wrow = 0; wcol = 16 * wave;
            tr_eff = (mrem + 15) >> 4; tc_eff = nrem > wcol ? 1 : 0;
        } else if (RELAYOUT && nrem <= 32) {     // 4 x 1 waves: block row `wave`, both block columns
            wrow = 16 * wave; wcol = 0;
            tr_eff = mrem > wrow ? 1 : 0; tc_eff = (nrem + 15) >> 4;
        } else {
            // 16 x 16 accumulator blocks of this wave that intersect the output (edge tiles): blocks outside are never
            // multiplied, so ragged sector sizes cost MFMA time at 16-granularity, not at tile granularity (wave-uniform).
            tr_eff = min(TR, max(0, (mrem - wrow + 15) >> 4)); tc_eff = min(TC, max(0, (nrem - wcol + 15) >> 4));
        }
        const bool wave_full = (tr_eff == TR) && (tc_eff == TC);
        unsigned lds_a, lds_b;
        {
            int tf = tid;
            asm volatile("" : "+v"(tf));
            const int l15 = tf & 15, l4 = (tf >> 4) & 3;
            lds_a = (unsigned)(size_t)&As[0][(wrow + l15) * AS_LD + l4];
            lds_b = (unsigned)(size_t)&Bs[0][l4 * BS_LD + wcol + l15];
        }
        double fa[2][TR], fb[2][TC];

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
        const int pend = g.prod_end;
        int p = tl.pad >= 0 ? tl.pad : pend;       // GTile::pad: the group's first GEMM product, -1 if it has none
        int k0 = 0;
        bool have = p < pend;
        if (!pre && have) {                        // first tile of the workgroup (or the tile before had no GEMM product)
            GG_LOADER_TILE(tl, g);
            GG_PRODUCT(p);
            GG_STAMP(1);
            GG_GLOAD(0);
        }
        if (have) GG_LSTORE(0);                    // (pre: the registers were filled beside the previous tile's epilogue)
        pre = false;
        __syncthreads();
        GG_STAMP(2);
        GG_CLK0();
        if (have) GG_FRAG(0, 0, 0);
        unsigned gmask = 0;                        // bit q: accumulator block q = mi * TC + ni of this wave intersects the output
#pragma unroll
        for (int q = 0; q < TR * TC; ++q) gmask |= ((q / TC) < tr_eff && (q % TC) < tc_eff) ? (1u << q) : 0u;
gmask = __builtin_amdgcn_readfirstlane(gmask);
// (the guard is TWO SCALAR instructions inside one asm statement: written as `if (gmask & bit)` the compiler keeps the sixteen tests as
//  lane masks and inverts them through the vector ALU -- ~24 v_cndmask / v_cmp per k-step, which the MFMA pipe cannot overlap)
#define GG_ALL(ACC, FA, FB, Q) ACC = __builtin_amdgcn_mfma_f64_16x16x4f64(FA, FB, ACC, 0, 0, 0)
#define GG_BIT(ACC, FA, FB, Q)                                                                \
        asm volatile("s_bitcmp1_b32 %3, %4\n\ts_cbranch_scc0 1f\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n1:" : "+v"(ACC) : "v"(FA), "v"(FB), "s"(gmask), "n"(Q) : "scc")
        if (TR * TC > 6) { GG_STREAM(GG_ALL) }
        else { GG_STREAM(GG_BIT) }
        GG_STAMP(3);
        GG_CLK1();
        // the staging registers hold nothing that is needed any more (the next tile's loads below set all of them; so does the
        // prologue of a tile that starts cold) -- said explicitly, because they are carried round the tile loop
#pragma unroll
        for (int s = 0; s < NA; ++s) asm volatile("" : "=v"(ra[s]));
#pragma unroll
        for (int s = 0; s < NB; ++s) asm volatile("" : "=v"(rb[s]));
        // this workgroup's next tile (claimed at the start of this one) and its descriptors: group and first product both depend
        // on the tile record only (GTile::pad = its first GEMM product); they travel while the scaled copies below are added
        int t_n = ntiles;
        if (dyn) {
            if (tid == 0) sh_next = atomicAdd(&ctr[xcd], 1);
            __syncthreads();
            t_n = xcd + 8 * (__builtin_amdgcn_readfirstlane(sh_next) + qbase);      // entries below qbase are the starting tiles
        }
        const GTile tl_n = t_n < ntiles ? kload(tiles, t_n) : tl_none;               // (padding ends a queue)
        const bool nx = tl_n.group >= 0;
        GGroup g_n = g;
        GProd pr_n = GProd{nullptr, nullptr, 0, 0, 0, 0, 0.0};
        if (nx) g_n = kload(groups, tl_n.group);
        if (nx && tl_n.pad >= 0) pr_n = kload(prods, tl_n.pad);

        // ---- scaled-copy products (identity operator cells): acc += alpha * S[tile] ------------------------
        // (after the GEMM stream: its value registers then do not meet the stream's operand / fragment registers)
        // A scaled copy is a descriptor load followed by a tile load; one product at a time the chain of both latencies (~2.5 us)
        // is paid per product.  The copies of a tile therefore form a pipeline of their own: while product q is added, the
        // descriptor of q + 1 is already in SGPRs and its tile loads are issued one block row at a time into the registers the
        // addition has just freed -- 16 values in flight per lane, not 32 (the two-products-at-once form of round 2 needed 64
        // value registers; inside the resident loop that pushed loop-invariant lane constants of the k-step into scratch).
        // Everything per-lane below derives from `tq`, an opaque copy of the thread index taken HERE: with the plain lane constants
        // the compiler computed the clamped rows / offsets of these phases ahead of the k-step stream and kept them -- 35 registers
        // -- in scratch across it.  Addresses are "wave-uniform 64-bit base + 32-bit lane offset" (as in the loaders).
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int el15 = tq & 15, el4 = (tq >> 4) & 3;
        // (the loader's offset registers are carried round the tile loop; their old contents are dead here -- every path sets them
        //  again before they are used -- which the compiler cannot see: overwritten with a lane-dependent dummy, they are not kept
        //  alive, i.e. in scratch, through the scaled copies)
#pragma unroll
        for (int s = 0; s < NA; ++s) aoff[s] = (unsigned)tq;
#pragma unroll
        for (int s = 0; s < NB; ++s) boff[s] = (unsigned)tq;
        const int erow = wrow + el4, ecol = wcol + el15;               // lane's first row / column inside the tile
        int q = g.prod_begin;
        const int axpy_end = q + g.n_axpy;
        const bool full_tile = mrem == BM && nrem == BN;
        if (TR * TC <= 4 && full_tile && q < axpy_end) {
            // full tile: row (mi, r) of the lane is a wave-uniform stride away from its first row, so the sixteen loads of a product
            // share ONE lane offset (plus an immediate for the second block column) off sixteen scalar bases
            double v[TR][TC][4];
            GProd pa = kload(prods, q);
#define GG_SLOAD(MI, PR)                                                                      \
            {                                                                                 \
                const unsigned lo_ = ((unsigned)erow * (unsigned)(PR).ldb + (unsigned)ecol) * 8u;   \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                               \
                    gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + (MI) * 16 + 4 * r) * (PR).ldb + n0);   \
                    _Pragma("unroll") for (int ni = 0; ni < TC; ++ni) v[MI][ni][r] = *(gptr)(S_ + lo_ + ni * 128); \
                }                                                                             \
            }
#pragma unroll
            for (int mi = 0; mi < TR; ++mi) GG_SLOAD(mi, pa);
            for (; q < axpy_end; ++q) {
                const bool more = q + 1 < axpy_end;
                const GProd pb = more ? kload(prods, q + 1) : pa;
#pragma unroll
                for (int mi = 0; mi < TR; ++mi) {
#pragma unroll
                    for (int ni = 0; ni < TC; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mi][ni][r] += pa.alpha * v[mi][ni][r];
                    if (more) GG_SLOAD(mi, pb);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pa = pb;
            }
#undef GG_SLOAD
        }
        for (; q < axpy_end; ++q) {
            const GProd pr = kload(prods, q);
            gbptr S = (gbptr)(pr.B + (size_t)m0 * pr.ldb + n0);
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int ni = 0; ni < TC; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = erow + mi * 16 + 4 * r, col = ecol + ni * 16;
                        if (row < mrem && col < nrem && mi < tr_eff && ni < tc_eff) acc[mi][ni][r] += pr.alpha * *(gptr)(S + ((unsigned)row * (unsigned)pr.ldb + (unsigned)col) * 8u);
                    }
        }
        GG_STAMP(4);

        // ---- the next tile's first operands are fetched beside this tile's output stores ------------------------
        if (nx && tl_n.pad >= 0) {
            GG_LOADER_TILE(tl_n, g_n);
            GG_PRODUCT_SET(pr_n);
            GG_GLOAD(0);
            pre = true;
        }

        // ---- epilogue ---------------------------------------------------------------------------------------
        if (full_tile && !g.accumulate) {
            const unsigned lo = ((unsigned)erow * (unsigned)g.ldc + (unsigned)ecol) * 8u;
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    char __attribute__((address_space(1)))* C = (char __attribute__((address_space(1)))*)(g.C + (size_t)(m0 + mi * 16 + 4 * r) * g.ldc + n0);
#pragma unroll
                    for (int ni = 0; ni < TC; ++ni) *(gwptr)(C + lo + ni * 128) = acc[mi][ni][r];
                }
        } else {
            char __attribute__((address_space(1)))* C = (char __attribute__((address_space(1)))*)(g.C + (size_t)m0 * g.ldc + n0);
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = erow + mi * 16 + 4 * r;
                    const unsigned ro = (unsigned)row * (unsigned)g.ldc;
#pragma unroll
                    for (int ni = 0; ni < TC; ++ni) {
                        const int col = ecol + ni * 16;
                        if (row < mrem && col < nrem && mi < tr_eff && ni < tc_eff) {
                            gwptr c = (gwptr)(C + (ro + (unsigned)col) * 8u);
                            *c = g.accumulate ? (*c + acc[mi][ni][r]) : acc[mi][ni][r];
                        }
                    }
                }
        }
        GG_STAMP(5);
        GG_STAMP_ID();
        if (!nx) break;
        t = t_n; tl = tl_n; g = g_n;
    }
    if (dyn && tid == 0 && atomicAdd(&ctr[8], 1) == G - 1) { for (int i = 0; i < 9; ++i) ctr[i] = 0; }   // the last one re-arms the counters
#undef GG_STREAM
#undef GG_STEP
#undef GG_FRAG
#undef GG_FRAG_WAIT
#undef GG_MFMA
#undef GG_PRODUCT
#undef GG_PRODUCT_SET
#undef GG_LOADER_TILE
#undef GG_GLOAD
#undef GG_LSTORE
}

int ggemm_cluster()
{
    static const int c = [] { const char* e = getenv("DMRGX_CLUSTER"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 16 ? v : 8; }();
    return c;
}

bool ggemm_use_big_tiles()
{
    const char* e = getenv("DMRGX_TILES");
    return e && std::string(e) == "mixed";
}

// Resident workgroups of one launch: every workgroup slot of the chip (4 per CU for the 64 x 64 kernel, 2 for the 128 x 128 one).
int ggemm_slots(int unit)
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        (void)hipGetLastError();
        return (n / 8) * 8 > 0 ? (n / 8) * 8 : 8;
    }();
    static const int per_cu = [] { const char* e = getenv("DMRGX_SLOTS"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 4 ? v : 4; }();   // experiment knob
    return (unit == 2 ? (per_cu + 1) / 2 : per_cu) * cus;
}

// Cost of a tile in units of a full tile's k-step, for the balance of the per-workgroup lists (tools/tile_trace.sh, cfg4real: a full
// tile's k-step takes 1.95 us with four streams on the CU, a thin edge tile's 1.0-1.4 us; a tile costs ~6 us = 3 k-steps
// outside its stream, a scaled copy ~1.2 us).
static inline int64_t tile_cost(const GTile& t) { return 4 * (int64_t)t.pad + 12; }

void ggemm_schedule_core(std::vector<GTile>& tiles, const std::vector<int32_t>& first_gemm, int unit)
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
        while (j < tiles.size() && tiles[j].group == tiles[i].group && tiles[j].tm / W == cm && tiles[j].tn / W == cn && j - i < 256) { c += tile_cost(tiles[j]); ++j; }
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
    std::vector<std::vector<GTile>> bins(NX);
    size_t maxbin = 0;
    for (int x = 0; x < NX; ++x) {
        // (coarse buckets of 16 k-steps, stable: clusters of similar tile length keep the LPT order, which keeps the clusters
        //  of one group -- same operands -- close together)
        std::stable_sort(binc[x].begin(), binc[x].end(), [&](const Cl& a, const Cl& b) { return (tiles[a.begin].pad >> 4) > (tiles[b.begin].pad >> 4); });
        for (const Cl& c : binc[x]) for (size_t t = c.begin; t < c.end; ++t) bins[x].push_back(tiles[t]);
        maxbin = std::max(maxbin, bins[x].size());
    }
    // eight interleaved per-XCD queues, padded with group = -1: workgroup b starts on entry b, the resident workgroups of an XCD
    // then claim the rest of its queue in order (a cluster's tiles next to each other, so that they run at the same time and meet in
    // the L2); see the kernel
    std::vector<GTile> out(maxbin * NX, GTile{-1, 0, 0, 0});
    for (int x = 0; x < NX; ++x) for (size_t i = 0; i < bins[x].size(); ++i) out[i * NX + x] = bins[x][i];
    // device meaning of GTile::pad: the group's first GEMM product (both descriptor loads of a tile then depend on the tile record only)
    for (GTile& t : out) t.pad = t.group >= 0 ? first_gemm[(size_t)t.group] : -1;
    tiles.swap(out);
}

#ifdef DMRGX_TILE_TRACE
static unsigned long long* g_trace_buf = nullptr;      // device buffer, 8 stamps per tile-list entry, launches appended one after the other
static size_t g_trace_cap = 0, g_trace_used = 0;
extern "C" void dmrgx_debug_tile_trace(void* dev_buf, size_t capacity_words) { g_trace_buf = (unsigned long long*)dev_buf; g_trace_cap = capacity_words; g_trace_used = 0; }
extern "C" size_t dmrgx_debug_tile_trace_used() { return g_trace_used; }
#define GG_TRACE_ARG , trace_
#else
#define GG_TRACE_ARG
#endif

dmrgx_status ggemm_launch(const GTile* d_tiles, const GGroup* d_groups, const GProd* d_prods, int32_t ntiles, hipStream_t st, int big)
{
    if (ntiles <= 0) return DMRGX_OK;
#ifdef DMRGX_TILE_TRACE
    unsigned long long* trace_ = nullptr;
    if (g_trace_buf && g_trace_used + (size_t)ntiles * 8 <= g_trace_cap) { trace_ = g_trace_buf + g_trace_used; g_trace_used += (size_t)ntiles * 8; }
#endif
    // resident workgroups: one per workgroup slot of the chip (or per entry, if there are fewer); a multiple of 8 whenever the
    // entries outnumber them, so that workgroup b sits on XCD b & 7 like the entries it starts on
    const unsigned grid = (unsigned)std::min(ntiles, ggemm_slots(big ? 2 : 1));
    // claim counters of this launch: 16 ints out of a ring (zero when the launch starts: the last workgroup of the launch that used
    // them before re-armed them; launches that could overlap -- other streams -- are 256 launches apart in the ring)
    static int* ring = nullptr;
    static std::atomic<unsigned> next{0};
    static std::once_flag once;
    static hipError_t ring_err = hipSuccess;
    std::call_once(once, [] {
        ring_err = hipMalloc((void**)&ring, 256 * 16 * sizeof(int));
        if (ring_err == hipSuccess) ring_err = hipMemset(ring, 0, 256 * 16 * sizeof(int));
    });
    DMRGX_HIP(ring_err);
    int* ctr = ring + 16 * (next.fetch_add(1) % 256u);
    if (big) hipLaunchKernelGGL((ggemm_kernel<4, 2, 2, 4>), dim3(grid), dim3(512), 0, st, d_tiles, d_groups, d_prods, ntiles, ctr GG_TRACE_ARG);
    else hipLaunchKernelGGL((ggemm_kernel<2, 2, 2, 2>), dim3(grid), dim3(256), 0, st, d_tiles, d_groups, d_prods, ntiles, ctr GG_TRACE_ARG);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

}  // namespace dmrgx
