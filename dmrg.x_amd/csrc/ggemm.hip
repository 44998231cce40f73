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
#include <map>
#include <mutex>
#include <utility>

namespace dmrgx {

typedef double d4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DDMRGX_TILE_TRACE, tools/tile_trace.sh; never compiled into the product library): every workgroup stamps
// the 100 MHz real-time counter at the milestones of its tile into a buffer of its own -- no output value depends on a stamp.
#ifdef DMRGX_TILE_TRACE
#define GG_TRACE_PARAM , unsigned long long* __restrict__ trace
#define GG_STAMP(I) { if (trace && threadIdx.x == 0) trace[(size_t)t * 8 + (I)] = __builtin_amdgcn_s_memrealtime(); }
#define GG_STAMP_ID() { if (trace && threadIdx.x == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); trace[(size_t)t * 8 + 7] = ((unsigned long long)xcc << 32) | hw; } }
#define GG_TRACE_FWD , trace
#define GG_CLK0() unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
#define GG_CLK1() { if (trace && threadIdx.x == 0) trace[(size_t)t * 8 + 6] = __builtin_amdgcn_s_memtime() - clk0_; }   /* shader cycles of the k-step stream */
#else
#define GG_TRACE_PARAM
#define GG_TRACE_FWD
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

// Workgroup -> tiles: a launch has G = min(ntiles, workgroup slots of the chip) RESIDENT workgroups.  The scheduled list is eight
// interleaved per-XCD queues (entry i belongs to XCD i & 7: blocks b, b+8, b+16, ... are the workgroups the dispatcher deals to one
// XCD; ggemm_schedule balances the queues and keeps a cluster's tiles next to each other); workgroup b starts on entry b and then
// claims the further entries of its XCD's queue, in order, with an atomic counter.
//
// Why resident workgroups (round 4; tools/tile_trace.sh on cfg4real, profiles/r04_tile_trace_before.txt): with one workgroup per
// tile a stage-1 launch kept 3.60 of 4 workgroup slots per CU occupied (7.5 % of the slot time lay between the end of one
// workgroup and the start of the next on the same CU: the dispatcher refills slots in order, median 1.8 us, p90 16.6 us) and a
// resident workgroup spent 13.6 % of its time outside the k-step stream (descriptor chain 1.2 us, first operands 1.6 us, scaled
// copies 2.4 us, stores 3.0 us per 52 us tile), so only 3.12 workgroups per CU were feeding the MFMA pipe.  A resident workgroup
// (a) is never re-dispatched, (b) claims its next tile while its loader has already left the current one (the last two k-steps
// hide the atomic), (c) fetches the next tile's first two k-steps beside the current tile's output stores.  What the measurements
// on the way taught (profiles/r04_tile_trace_*.txt, DESIGN.md K1 "Round 4"):
//   * the claim must be LATE.  Claimed at the start of the current tile (a whole tile ahead), the tiles of an 8 x 8 cluster started
//     over 77-120 us instead of 30 us, i.e. at different positions of the shared panels, and the L2 hit rate fell from 82 % to 66 %
//     (FETCH_SIZE 2.1 x): the in-order dispatcher of the old scheme had kept them together for free;
//   * spilled registers thrash the L2: 27 spilled dwords are 7 KB per wave, 3.5 MB per XCD -- the size of its L2.  This kernel
//     spills nothing (see the staging registers and GG_LANES);
//   * the chip holds ~2.05 GHz under this load: cycles saved in the stream come back partly as a lower clock.
//
// TR x TC v_mfma_f64_16x16x4 accumulators per wave, WR x WC waves per workgroup: tile (16 TR WR) x (16 TC WC).
//   <2,2,2,2>:  64 x  64 tile, 256 threads, 4 workgroups per CU -- ragged remainders and small sectors
//   <4,2,2,4>: 128 x 128 tile, 512 threads, 2 workgroups per CU -- same 4 waves per SIMD, half the L2->LDS bytes
//                                                                 per flop, 64 accumulator registers per wave
//   DEEP: the loader runs two k-steps ahead through hand-managed staging registers (see GG_GLOAD_D); else one k-step ahead through
//         ordinary variables (the 128 x 128 shape has no 32 registers to spare for a second staging set)
template <int TR, int TC, int WR, int WC, bool DEEP>
__device__ __forceinline__ void
ggemm_body(const GTile* __restrict__ tiles_, const GGroup* __restrict__ groups_, const GProd* __restrict__ prods_, int ntiles, int* __restrict__ ctr GG_TRACE_PARAM)
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
    // LDS is DYNAMIC (size passed at launch: ggemm_lds_bytes): with a static size the compiler sees that LDS holds the kernel to
    // four waves per SIMD anyway and hands the register allocator all 128 registers, staging registers included
    extern __shared__ double gg_smem[];
    double (*As)[BM * AS_LD] = (double (*)[BM * AS_LD])gg_smem;
    double (*Bs)[BK * BS_LD] = (double (*)[BK * BS_LD])(gg_smem + 2 * BM * AS_LD);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    // Work distribution: the list is eight interleaved per-XCD queues (entry i belongs to XCD i & 7; the host balances the queues).
    // Workgroup b starts on entry b and then claims the further entries of its XCD's queue, in order, with an atomic counter
    // (ctr[0..7]: this launch's own block of counters, zeroed on the launch stream before the launch -- ggemm_launch; nothing is
    // re-armed by the kernel, so a launch that dies leaves nothing behind that a later one could trip over).  The claim for the NEXT
    // tile is issued when the loader leaves the current one and read after its k-step stream, so its latency is never waited for.
    int& sh_next = *(int*)(gg_smem + 2 * BM * AS_LD + 2 * BK * BS_LD);
    const int G = gridDim.x;
    const bool dyn = ntiles > G;                   // (else every entry has its own workgroup)
    const int xcd = blockIdx.x & 7, qbase = G >> 3;
    int t = blockIdx.x;
    GTile tl = kload(tiles, t);
    if (tl.group < 0) return;                       // padding: this queue is shorter than its share of starting tiles
    GGroup g = kload(groups, tl.group);
    const GTile tl_none = GTile{-1, 0, 0, 0};
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
    // Two staging register sets: the loader runs TWO k-steps ahead of the multiplication (set j & 1 holds k-step j between its
    // loads and its store into LDS buffer j & 1).  One step ahead -- rounds 1-3 -- a tile whose operands come from beyond the L2
    // (the first tile of a cluster to touch a panel: about one tile in six, tools/tile_trace.sh) waits for every one of its loads:
    // its k-steps took 5 300-6 000 cycles instead of 3 600.  The loads and their waits are written by hand (asm loads, counted
    // s_waitcnt vmcnt): the compiler waits for vmcnt(0) on loads that are in flight across a loop back edge.
    double rc_a[NA], rc_b[NB];                     // (!DEEP) the one staging set
    const double *cA = nullptr, *cB = nullptr;     // loader's current product, wave-uniform -> SGPRs
    int clda = 0, cldb = 0, cK = 0;
    int lp = 0, lk = 0, lpend = 0;                 // loader position: product, k offset of the next k-step to fetch, end of the tile's products
    bool lmore = false;                            // a k-step is left to fetch at (lp, lk)
    int kz[2] = {0, 0};                            // K edge of the k-step in a register set: > 0 columns below kz, < 0 columns from -kz on are zeroed
    int npre = 0;                                  // k-steps of tile `tl` already fetched into the register sets (beside the previous tile's epilogue)
    int vm_after = 0;                              // vector-memory operations issued after the loads of register set 0 (for its counted wait)

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
// DEEP staging registers: v96 .. v127 belong to the hand-written statements below and to nothing else.  The 64 x 64 kernel is compiled
// for five waves per SIMD, so the compiler allocates v0 .. v95 only; the statements name the staging registers literally and list
// them as clobbers (which is also what makes the kernel's register count 128).  No C++ value ever lives in them: between an asm
// load and the asm wait a compiler would consider such a value arrived and might copy it (live-range split, loop-head phi) while it is
// still in flight.  tools/check_staging_regs.py scans the ISA of every build: only asm statements may touch v96 .. v127.
#define GG_RA_0_0 "v[96:97]"
#define GG_CA_0_0 "v96", "v97"
#define GG_RA_0_1 "v[98:99]"
#define GG_CA_0_1 "v98", "v99"
#define GG_RA_0_2 "v[100:101]"
#define GG_CA_0_2 "v100", "v101"
#define GG_RA_0_3 "v[102:103]"
#define GG_CA_0_3 "v102", "v103"
#define GG_RB_0_0 "v[104:105]"
#define GG_CB_0_0 "v104", "v105"
#define GG_RB_0_1 "v[106:107]"
#define GG_CB_0_1 "v106", "v107"
#define GG_RB_0_2 "v[108:109]"
#define GG_CB_0_2 "v108", "v109"
#define GG_RB_0_3 "v[110:111]"
#define GG_CB_0_3 "v110", "v111"
#define GG_RA_1_0 "v[112:113]"
#define GG_CA_1_0 "v112", "v113"
#define GG_RA_1_1 "v[114:115]"
#define GG_CA_1_1 "v114", "v115"
#define GG_RA_1_2 "v[116:117]"
#define GG_CA_1_2 "v116", "v117"
#define GG_RA_1_3 "v[118:119]"
#define GG_CA_1_3 "v118", "v119"
#define GG_RB_1_0 "v[120:121]"
#define GG_CB_1_0 "v120", "v121"
#define GG_RB_1_1 "v[122:123]"
#define GG_CB_1_1 "v122", "v123"
#define GG_RB_1_2 "v[124:125]"
#define GG_CB_1_2 "v124", "v125"
#define GG_RB_1_3 "v[126:127]"
#define GG_CB_1_3 "v126", "v127"
#define GG_ZT_0_0 "v_cndmask_b32_e64 v96, v96, 0, %0\n\tv_cndmask_b32_e64 v97, v97, 0, %0"
#define GG_ZT_0_1 "v_cndmask_b32_e64 v98, v98, 0, %0\n\tv_cndmask_b32_e64 v99, v99, 0, %0"
#define GG_ZT_0_2 "v_cndmask_b32_e64 v100, v100, 0, %0\n\tv_cndmask_b32_e64 v101, v101, 0, %0"
#define GG_ZT_0_3 "v_cndmask_b32_e64 v102, v102, 0, %0\n\tv_cndmask_b32_e64 v103, v103, 0, %0"
#define GG_ZT_1_0 "v_cndmask_b32_e64 v112, v112, 0, %0\n\tv_cndmask_b32_e64 v113, v113, 0, %0"
#define GG_ZT_1_1 "v_cndmask_b32_e64 v114, v114, 0, %0\n\tv_cndmask_b32_e64 v115, v115, 0, %0"
#define GG_ZT_1_2 "v_cndmask_b32_e64 v116, v116, 0, %0\n\tv_cndmask_b32_e64 v117, v117, 0, %0"
#define GG_ZT_1_3 "v_cndmask_b32_e64 v118, v118, 0, %0\n\tv_cndmask_b32_e64 v119, v119, 0, %0"
#define GG_LD1(OP, SET, S, OFFV, BASE, BREG) asm volatile("global_load_dwordx2 " GG_R##OP##_##SET##_##S ", %0, %1" : : "v"(OFFV), BREG(BASE) : "memory", GG_C##OP##_##SET##_##S)
#define GG_ST1(OP, SET, S, ADDR, OFFS) asm volatile("ds_write_b64 %0, " GG_R##OP##_##SET##_##S " offset:%1" : : "v"(ADDR), "i"(OFFS) : "memory")
#define GG_Z1(SET, S) asm volatile(GG_ZT_##SET##_##S : : "s"(zm_) : GG_CA_##SET##_##S)
// position of the loader after the k-step at (lp, lk)
#define GG_LOADER_NEXT()                                                                      \
    {                                                                                         \
        lk += BK;                                                                             \
        if (lk >= cK) { ++lp; lk = 0; if (lp < lpend) GG_PRODUCT(lp); }                       \
        lmore = lp < lpend;                                                                   \
    }
#define GG_LOADER_BASES()                                                                     \
        const int klast_ = cK - 1 - lk;                                                       \
        const int back_ = (klast_ < BK - 1 && cK >= BK) ? BK - 1 - klast_ : 0;                \
        const int kz_ = klast_ < BK - 1 ? (cK >= BK ? back_ : -(klast_ + 1)) : 0;             \
        gbptr A_ = (gbptr)(cA + (size_t)lm0 * clda + (lk - back_));                           \
        gbptr B_ = (gbptr)(cB + (size_t)(lk - back_) * cldb + ln0);
// (DEEP) fetch the k-step at the loader's position into register set SET (8 asm loads, nothing waits), then move the loader on
#define GG_GLOAD_D(SET)                                                                       \
    {                                                                                         \
        static_assert(!DEEP || (NA == 4 && NB == 4), "staging register table");               \
        GG_LOADER_BASES();                                                                    \
        kz[SET] = kz_;                                                                        \
        GG_LD1(A, SET, 0, aoff[0], A_, "{s[96:97]}"); GG_LD1(A, SET, 1, aoff[1 % NA], A_, "{s[96:97]}");   \
        GG_LD1(A, SET, 2, aoff[2 % NA], A_, "{s[96:97]}"); GG_LD1(A, SET, 3, aoff[3 % NA], A_, "{s[96:97]}");   \
        GG_LD1(B, SET, 0, boff[0], B_, "{s[98:99]}"); GG_LD1(B, SET, 1, boff[1 % NB], B_, "{s[98:99]}");   \
        GG_LD1(B, SET, 2, boff[2 % NB], B_, "{s[98:99]}"); GG_LD1(B, SET, 3, boff[3 % NB], B_, "{s[98:99]}");   \
        GG_LOADER_NEXT();                                                                     \
    }
// all but the N youngest vector-memory operations of the wave have completed
#define GG_VMWAIT(N) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory")
#define GG_LSTORE_D(BUF, SET)                                                                 \
    {                                                                                         \
        if (kz[SET] != 0) {                                                                   \
            GG_LANES();                                                                       \
            const unsigned long long zm_ = __builtin_amdgcn_ballot_w64(kz[SET] > 0 ? a_k < kz[SET] : a_k >= -kz[SET]);   \
            GG_Z1(SET, 0); GG_Z1(SET, 1); GG_Z1(SET, 2); GG_Z1(SET, 3);                       \
        }                                                                                     \
        GG_ST1(A, SET, 0, st_a, 8 * ((BUF) * BM * AS_LD + AROWS * 0 * AS_LD)); GG_ST1(A, SET, 1, st_a, 8 * ((BUF) * BM * AS_LD + AROWS * 1 * AS_LD)); \
        GG_ST1(A, SET, 2, st_a, 8 * ((BUF) * BM * AS_LD + AROWS * 2 * AS_LD)); GG_ST1(A, SET, 3, st_a, 8 * ((BUF) * BM * AS_LD + AROWS * 3 * AS_LD)); \
        GG_ST1(B, SET, 0, st_b, 8 * ((BUF) * BK * BS_LD + BROWS * 0 * BS_LD)); GG_ST1(B, SET, 1, st_b, 8 * ((BUF) * BK * BS_LD + BROWS * 1 * BS_LD)); \
        GG_ST1(B, SET, 2, st_b, 8 * ((BUF) * BK * BS_LD + BROWS * 2 * BS_LD)); GG_ST1(B, SET, 3, st_b, 8 * ((BUF) * BK * BS_LD + BROWS * 3 * BS_LD)); \
    }
// (!DEEP) the same through ordinary variables: one set, the compiler places the waits
#define GG_GLOAD_C()                                                                          \
    {                                                                                         \
        GG_LOADER_BASES();                                                                    \
        kz[0] = kz_;                                                                          \
        _Pragma("unroll") for (int s = 0; s < NA; ++s) rc_a[s] = *(gptr)(A_ + aoff[s]);       \
        _Pragma("unroll") for (int s = 0; s < NB; ++s) rc_b[s] = *(gptr)(B_ + boff[s]);       \
        GG_LOADER_NEXT();                                                                     \
    }
#define GG_LSTORE_C(BUF)                                                                      \
    {                                                                                         \
        if (kz[0] != 0) {                                                                     \
            GG_LANES();                                                                       \
            const bool z_ = kz[0] > 0 ? a_k < kz[0] : a_k >= -kz[0];                          \
            _Pragma("unroll") for (int s = 0; s < NA; ++s) rc_a[s] = z_ ? 0.0 : rc_a[s];      \
        }                                                                                     \
        _Pragma("unroll") for (int s = 0; s < NA; ++s)                                        \
            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(st_a), "v"(rc_a[s]), "i"(8 * ((BUF) * BM * AS_LD + AROWS * s * AS_LD)) : "memory"); \
        _Pragma("unroll") for (int s = 0; s < NB; ++s)                                        \
            asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(st_b), "v"(rc_b[s]), "i"(8 * ((BUF) * BK * BS_LD + BROWS * s * BS_LD)) : "memory"); \
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
// The claim of the workgroup's next tile: one returning atomic add by lane 0 of wave 0, under a WAVE-UNIFORM branch and with the
// exec mask set by hand (a divergent `if (tid == 0)` inside the k-step loop makes the compiler treat the loader's scalar state as
// divergent).  The value lands in lane 0 of `claim`; nothing waits for it here.
// `claim` is the one C++ variable of this kernel that holds a value in flight (the staging registers are all taken until the stream
// ends): it is read-write for the statement ("+v": one register from its initialisation to the wait, no fresh definition the
// allocator could place elsewhere) and tools/check_staging_regs.py verifies on every build that no compiler-generated instruction
// touches that register between a claim and the s_waitcnt vmcnt(0) that precedes its first use.
#define GG_CLAIM()                                                                            \
    asm volatile("s_mov_b64 s[94:95], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_mov_b64 exec, s[94:95]" \
                 : "+v"(claim) : "v"(0u), "v"(1u), "{s[92:93]}"(ctr + xcd) : "memory", "s94", "s95")
// (DEEP) k-step j on LDS buffer BUF = j & 1.  On entry register set BUF is free (k-step j was stored from it), set OTH = BUF ^ 1 holds
// k-step j + 1 if there is one (`have1`), the loader stands at k-step j + 2.
#define GG_STEP_D(BUF, OTH, GUARD)                                                            \
    {                                                                                         \
        constexpr int NM = TR * TC;                                                           \
        GG_FRAG_WAIT(0);                                                                      \
        GG_MFMA(0, GUARD, 0, 1);                                                              \
        const bool ld_ = lmore;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, 2);                                                              \
        if (ld_) {                                                                            \
            GG_GLOAD_D(BUF);                                                                  \
            if (dyn && !lmore && wave == 0) GG_CLAIM();   /* the loader has left the tile: its last two k-steps hide the claim */ \
        }                                                                                     \
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
        if (have1) {                                                                          \
            if (ld_ && lmore) { GG_VMWAIT(NA + NB); } else { GG_VMWAIT(0); }   /* (a claim, if issued, is waited for too) */ \
            GG_LSTORE_D(OTH, OTH);                                                            \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, NM - 1);                                                         \
        GG_FRAG(1, BUF, 12);                                                                  \
        GG_MFMA(0, GUARD, NM - 1, NM);                                                        \
        GG_FRAG_WAIT(1);                                                                      \
        __syncthreads();                                                                      \
        if (have1) GG_FRAG(0, OTH, 0);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(1, GUARD, 0, NM);                                                             \
        have = have1; have1 = ld_;                                                            \
    }
// (!DEEP) the loader one k-step ahead: k-step j + 1 is fetched in k-group 0 and stored in k-group 2 of k-step j
#define GG_STEP_C(BUF, OTH, GUARD)                                                            \
    {                                                                                         \
        constexpr int NM = TR * TC;                                                           \
        GG_FRAG_WAIT(0);                                                                      \
        GG_MFMA(0, GUARD, 0, 1);                                                              \
        const bool ld_ = lmore;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, 2);                                                              \
        if (ld_) {                                                                            \
            GG_GLOAD_C();                                                                     \
            if (dyn && !lmore && wave == 0) GG_CLAIM();                                       \
        }                                                                                     \
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
        if (ld_) GG_LSTORE_C(OTH);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(0, GUARD, 1, NM - 1);                                                         \
        GG_FRAG(1, BUF, 12);                                                                  \
        GG_MFMA(0, GUARD, NM - 1, NM);                                                        \
        GG_FRAG_WAIT(1);                                                                      \
        __syncthreads();                                                                      \
        if (ld_) GG_FRAG(0, OTH, 0);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        GG_MFMA(1, GUARD, 0, NM);                                                             \
        have = ld_;                                                                           \
    }
// The k-step stream, instantiated twice: interior waves run it with every MFMA unconditional; waves on a ragged tile
// edge run a copy whose MFMAs are guarded by wave-uniform block bounds (blocks outside the output are never multiplied).
// Both copies execute the same barriers, so the waves of one workgroup may take different copies.  (One loop with a
// per-k-step choice makes the compiler merge the accumulators with v_mov copies that wait on the MFMA results.)
#define GG_STREAM(GUARD)                                                                      \
    while (have) {                                                                            \
        if (DEEP) { GG_STEP_D(0, 1, GUARD); } else { GG_STEP_C(0, 1, GUARD); }                \
        if (!have) break;                                                                     \
        if (DEEP) { GG_STEP_D(1, 0, GUARD); } else { GG_STEP_C(1, 0, GUARD); }                \
    }

    for (;;) {
        GG_STAMP(0);
        // per-thread byte offsets inside the loader's current product's panels: variables of ONE tile iteration (set before the stream
        // for this tile's loader, overwritten after it for the next tile's first loads), so that nothing is carried round the loop
        unsigned aoff[NA], boff[NB];
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
        const int p0 = tl.pad >= 0 ? tl.pad : pend;      // GTile::pad: the group's first GEMM product, -1 if it has none
        if (npre == 0 && p0 < pend) {              // first tile of the workgroup (or the tile before had no GEMM product): fetch its first k-step(s)
            GG_LOADER_TILE(tl, g);
            lp = p0; lk = 0; lpend = pend;
            GG_PRODUCT(lp);
            GG_STAMP(1);
            vm_after = 0;
            if (DEEP) { GG_GLOAD_D(0); npre = 1; if (lmore) { GG_GLOAD_D(1); npre = 2; vm_after = NA + NB; } }
            else { GG_GLOAD_C(); npre = 1; }
        } else if (lmore) {
            GG_PRODUCT(lp);                        // (first k-steps fetched beside the previous epilogue: the offsets of the loader's product again)
        } else {
#pragma unroll
            for (int s = 0; s < NA; ++s) aoff[s] = 0u;
#pragma unroll
            for (int s = 0; s < NB; ++s) boff[s] = 0u;
        }
        bool have = npre >= 1, have1 = npre >= 2;
        (void)have1;
        if (have) {
            if (DEEP) {
                // register set 0 -> LDS buffer 0, after a wait that leaves the younger operations in flight: the loads of set 1 and, if
                // the sets were filled beside the previous tile's epilogue, its output stores
                if (vm_after == NA + NB) { GG_VMWAIT(NA + NB); }
                else if (vm_after == NA + NB + TR * TC * 4) { GG_VMWAIT(NA + NB + TR * TC * 4); }
                else if (vm_after == TR * TC * 4) { GG_VMWAIT(TR * TC * 4); }
                else { GG_VMWAIT(0); }
                GG_LSTORE_D(0, 0);
            } else GG_LSTORE_C(0);
        }
        npre = 0;
        unsigned claim = 0xffffffffu;              // (lane 0 of wave 0) counter value claimed for the next tile
        if (dyn && have && !lmore && wave == 0) GG_CLAIM();      // (a tile so short that the stream below never fetches)
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
        for (int s = 0; s < NA; ++s) asm volatile("" : "=v"(rc_a[s]));
#pragma unroll
        for (int s = 0; s < NB; ++s) asm volatile("" : "=v"(rc_b[s]));
        // this workgroup's next tile (claimed at the start of this one) and its descriptors: group and first product both depend
        // on the tile record only (GTile::pad = its first GEMM product); they travel while the scaled copies below are added
        int t_n = ntiles;
        if (dyn) {
            // (the claim was issued by an asm statement: the compiler does not know that `claim` may still be in flight)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(claim) : : "memory");
            if (tid == 0) sh_next = claim != 0xffffffffu ? (int)claim : (int)atomicAdd(&ctr[xcd], 1);      // (claimed before the end of the stream)
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
        const int erow = wrow + el4, ecol = wcol + el15;               // lane's first row / column inside the tile
        int q = g.prod_begin;
        const int axpy_end = q + g.n_axpy;
        const bool full_tile = mrem == BM && nrem == BN;
        if (DEEP && full_tile && q < axpy_end) {
            // full tile: row (mi, r) of the lane is a wave-uniform stride away from its first row, so the sixteen loads of a product
            // share ONE lane offset (plus an immediate for the second block column) off eight scalar bases.  The sixteen values land
            // in the STAGING registers (v96 .. v127, idle between two streams; statements by hand as for the operands): as a C++ array
            // they were 32 of the compiler's 96 registers and pushed the accumulators' neighbours into scratch.
#define GG_SREG_0 "v[96:97]"
#define GG_SCLB_0 "v96", "v97"
#define GG_SREG_1 "v[98:99]"
#define GG_SCLB_1 "v98", "v99"
#define GG_SREG_2 "v[100:101]"
#define GG_SCLB_2 "v100", "v101"
#define GG_SREG_3 "v[102:103]"
#define GG_SCLB_3 "v102", "v103"
#define GG_SREG_4 "v[104:105]"
#define GG_SCLB_4 "v104", "v105"
#define GG_SREG_5 "v[106:107]"
#define GG_SCLB_5 "v106", "v107"
#define GG_SREG_6 "v[108:109]"
#define GG_SCLB_6 "v108", "v109"
#define GG_SREG_7 "v[110:111]"
#define GG_SCLB_7 "v110", "v111"
#define GG_SREG_8 "v[112:113]"
#define GG_SCLB_8 "v112", "v113"
#define GG_SREG_9 "v[114:115]"
#define GG_SCLB_9 "v114", "v115"
#define GG_SREG_10 "v[116:117]"
#define GG_SCLB_10 "v116", "v117"
#define GG_SREG_11 "v[118:119]"
#define GG_SCLB_11 "v118", "v119"
#define GG_SREG_12 "v[120:121]"
#define GG_SCLB_12 "v120", "v121"
#define GG_SREG_13 "v[122:123]"
#define GG_SCLB_13 "v122", "v123"
#define GG_SREG_14 "v[124:125]"
#define GG_SCLB_14 "v124", "v125"
#define GG_SREG_15 "v[126:127]"
#define GG_SCLB_15 "v126", "v127"
#define GG_SLD(U, OFFV, BASE, IMM) asm volatile("global_load_dwordx2 " GG_SREG_##U ", %0, %1 offset:" #IMM : : "v"(OFFV), "{s[96:97]}"(BASE) : "memory", GG_SCLB_##U)
#define GG_SFMA(U, EL, AL) { double t_ = EL; asm volatile("v_fma_f64 %0, %1, " GG_SREG_##U ", %0" : "+v"(t_) : "s"(AL)); EL = t_; }
#define GG_SLOAD0(PR)                                                                         \
            {                                                                                 \
                const unsigned lo_ = ((unsigned)erow * (unsigned)(PR).ldb + (unsigned)ecol) * 8u;   \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 0 + 4 * 0) * (PR).ldb + n0); GG_SLD(0, lo_, S_, 0); GG_SLD(1, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 0 + 4 * 1) * (PR).ldb + n0); GG_SLD(2, lo_, S_, 0); GG_SLD(3, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 0 + 4 * 2) * (PR).ldb + n0); GG_SLD(4, lo_, S_, 0); GG_SLD(5, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 0 + 4 * 3) * (PR).ldb + n0); GG_SLD(6, lo_, S_, 0); GG_SLD(7, lo_, S_, 128); } \
            }
#define GG_SLOAD1(PR)                                                                         \
            {                                                                                 \
                const unsigned lo_ = ((unsigned)erow * (unsigned)(PR).ldb + (unsigned)ecol) * 8u;   \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 16 + 4 * 0) * (PR).ldb + n0); GG_SLD(8, lo_, S_, 0); GG_SLD(9, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 16 + 4 * 1) * (PR).ldb + n0); GG_SLD(10, lo_, S_, 0); GG_SLD(11, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 16 + 4 * 2) * (PR).ldb + n0); GG_SLD(12, lo_, S_, 0); GG_SLD(13, lo_, S_, 128); } \
                { gbptr S_ = (gbptr)((PR).B + (size_t)(m0 + 16 + 4 * 3) * (PR).ldb + n0); GG_SLD(14, lo_, S_, 0); GG_SLD(15, lo_, S_, 128); } \
            }
#define GG_SFMA0(AL)                                                                          \
            {                                                                                 \
                GG_SFMA(0, acc[0][0][0], AL); \
                GG_SFMA(1, acc[0][1][0], AL); \
                GG_SFMA(2, acc[0][0][1], AL); \
                GG_SFMA(3, acc[0][1][1], AL); \
                GG_SFMA(4, acc[0][0][2], AL); \
                GG_SFMA(5, acc[0][1][2], AL); \
                GG_SFMA(6, acc[0][0][3], AL); \
                GG_SFMA(7, acc[0][1][3], AL); \
            }
#define GG_SFMA1(AL)                                                                          \
            {                                                                                 \
                GG_SFMA(8, acc[1][0][0], AL); \
                GG_SFMA(9, acc[1][1][0], AL); \
                GG_SFMA(10, acc[1][0][1], AL); \
                GG_SFMA(11, acc[1][1][1], AL); \
                GG_SFMA(12, acc[1][0][2], AL); \
                GG_SFMA(13, acc[1][1][2], AL); \
                GG_SFMA(14, acc[1][0][3], AL); \
                GG_SFMA(15, acc[1][1][3], AL); \
            }
            static_assert(!DEEP || (TR == 2 && TC == 2), "scaled-copy register table");
            GProd pa = kload(prods, q);
            GG_SLOAD0(pa); GG_SLOAD1(pa);
            for (; q < axpy_end; ++q) {
                const bool more = q + 1 < axpy_end;
                const GProd pb = more ? kload(prods, q + 1) : pa;
                const double al = pa.alpha;
                GG_VMWAIT(8);                       // block row 0 of product q has landed (block row 1: the 8 younger loads)
                GG_SFMA0(al);
                if (more) { GG_SLOAD0(pb); GG_VMWAIT(8); } else { GG_VMWAIT(0); }
                GG_SFMA1(al);
                if (more) GG_SLOAD1(pb);
                pa = pb;
            }
#undef GG_SLOAD0
#undef GG_SLOAD1
#undef GG_SFMA0
#undef GG_SFMA1
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
        vm_after = 0;
        if (nx && tl_n.pad >= 0) {
            GG_LOADER_TILE(tl_n, g_n);
            lp = tl_n.pad; lk = 0; lpend = g_n.prod_end;
            GG_PRODUCT_SET(pr_n);
            if (DEEP) { GG_GLOAD_D(0); npre = 1; if (lmore) { GG_GLOAD_D(1); npre = 2; vm_after = NA + NB; } }
            else { GG_GLOAD_C(); npre = 1; }
        }

        // ---- epilogue ---------------------------------------------------------------------------------------
        if (full_tile && !g.accumulate) {
            // TR * TC * 4 stores, each ONE asm statement: the next tile's counted wait (vm_after) relies on exactly this many vector-memory
            // operations being issued here -- fewer would release register set 0 before its loads have landed
            const unsigned lo = ((unsigned)erow * (unsigned)g.ldc + (unsigned)ecol) * 8u;
#pragma unroll
            for (int mi = 0; mi < TR; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    char __attribute__((address_space(1)))* C = (char __attribute__((address_space(1)))*)(g.C + (size_t)(m0 + mi * 16 + 4 * r) * g.ldc + n0);
#pragma unroll
                    for (int ni = 0; ni < TC; ++ni)
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" : : "v"(lo), "v"(acc[mi][ni][r]), "s"(C), "n"(ni * 128) : "memory");
                }
            vm_after += TR * TC * 4;
        } else {
            vm_after = -1;                        // (an unknown number of stores / loads: the next tile waits for all of them)
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
#undef GG_STREAM
#undef GG_STEP
#undef GG_FRAG
#undef GG_FRAG_WAIT
#undef GG_MFMA
#undef GG_PRODUCT
#undef GG_PRODUCT_SET
#undef GG_LOADER_TILE
#undef GG_GLOAD_D
#undef GG_GLOAD_C
#undef GG_LSTORE_D
#undef GG_LSTORE_C
}

// (waves_per_eu(5, 5) is how the compiler is held to v0 .. v95 -- 512 / 5 rounded down to the allocation granule; with the staging
//  registers named by the asm statements the kernel's register count is 128 and four waves per SIMD are resident, as intended.  clang's
//  amdgpu_num_vgpr attribute is ignored by this toolchain.)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
ggemm_kernel_64(const GTile* __restrict__ tiles, const GGroup* __restrict__ groups, const GProd* __restrict__ prods, int ntiles, int* __restrict__ ctr GG_TRACE_PARAM)
{
    ggemm_body<2, 2, 2, 2, true>(tiles, groups, prods, ntiles, ctr GG_TRACE_FWD);
}
__global__ void __launch_bounds__(512, 4)
ggemm_kernel_128(const GTile* __restrict__ tiles, const GGroup* __restrict__ groups, const GProd* __restrict__ prods, int ntiles, int* __restrict__ ctr GG_TRACE_PARAM)
{
    ggemm_body<4, 2, 2, 4, false>(tiles, groups, prods, ntiles, ctr GG_TRACE_FWD);
}


// Resident workgroups of one launch: every workgroup slot of the chip (4 per CU for the 64 x 64 kernel, 2 for the 128 x 128 one).
int ggemm_slots(int unit)
{
    // per device (a process may drive several: dmrgx_set_device), looked up once each
    static std::mutex mu;
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    int cus;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (cus_of[dev] == 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
            cus_of[dev] = (n / 8) * 8 > 0 ? (n / 8) * 8 : 8;
        }
        cus = cus_of[dev];
    }
    return (unit == 2 ? 2 : 4) * cus;
}

// Cost of a tile in units of a full tile's k-step, for the balance of the per-workgroup lists (tools/tile_trace.sh, cfg4real: a full
// tile's k-step takes 1.95 us with four streams on the CU, a thin edge tile's 1.0-1.4 us; a tile costs ~6 us = 3 k-steps
// outside its stream, a scaled copy ~1.2 us).
static inline int64_t tile_cost(const GTile& t) { return 4 * (int64_t)t.pad + 12; }

void ggemm_schedule_core(std::vector<GTile>& tiles, const std::vector<int32_t>& first_gemm, int unit)
{
    // (host cost matters: every grouped-GEMM user calls this once per list, ~40 lists in a sweep step, 30 000 tiles in the MatMult's --
    //  shifts instead of divisions, one pass per stage, the output written in place: profiles/r04_hostprof_glue_m2048.txt)
    constexpr int NX = 8;
    if (tiles.empty()) return;
    // A list that fits a quarter of the chip's workgroup slots is launched one workgroup per entry, all at once: neither the order nor the
    // balance of the eight queues matters, and at small m a sweep step schedules ~40 such lists (3 % of the host's time at m = 512,
    // profiles/r05_hostprof_m512.txt).  Only the device meaning of GTile::pad is filled in.
    if (tiles.size() <= (size_t)ggemm_slots(unit) / 4) {
        for (GTile& t : tiles) t.pad = first_gemm[(size_t)t.group];
        return;
    }
    struct Cl { uint32_t begin, end; int32_t len16; int64_t cost; };
    static_assert(GG_CLUSTER == 8, "cluster edge as a shift");
    const int sh = unit == 2 ? 4 : 3;                 // tiles of 128 x 128 carry tm / tn in units of 64: a cluster is 16 units wide
    static thread_local std::vector<Cl> cl;
    cl.clear();
    const size_t nt = tiles.size();
    for (size_t i = 0; i < nt;) {
        size_t j = i;
        int64_t c = 0;
        // a cluster = the consecutive tiles of one group inside one GG_CLUSTER x GG_CLUSTER block of its tile grid
        const int32_t grp = tiles[i].group, cm = tiles[i].tm >> sh, cn = tiles[i].tn >> sh;
        while (j < nt && tiles[j].group == grp && (tiles[j].tm >> sh) == cm && (tiles[j].tn >> sh) == cn && j - i < 256) { c += tile_cost(tiles[j]); ++j; }
        cl.push_back(Cl{(uint32_t)i, (uint32_t)j, tiles[i].pad >> 4, c});
        i = j;
    }
    std::stable_sort(cl.begin(), cl.end(), [](const Cl& a, const Cl& b) { return a.cost > b.cost; });
    // clusters go to the least-loaded XCD, heaviest first (LPT) ...
    static thread_local std::vector<Cl> binc[NX];
    int64_t load[NX] = {0};
    size_t count[NX] = {0};
    for (int x = 0; x < NX; ++x) binc[x].clear();
    for (const Cl& c : cl) {
        int best = 0;
        for (int x = 1; x < NX; ++x) if (load[x] < load[best]) best = x;
        load[best] += c.cost;
        count[best] += c.end - c.begin;
        binc[best].push_back(c);
    }
    size_t maxbin = 0;
    for (int x = 0; x < NX; ++x) maxbin = std::max(maxbin, count[x]);
    // eight interleaved per-XCD queues, padded with group = -1: workgroup b starts on entry b, the resident workgroups of an XCD
    // then claim the rest of its queue in order (a cluster's tiles next to each other, so that they run at the same time and meet in
    // the L2); see the kernel.  Device meaning of GTile::pad: the group's first GEMM product (both descriptor loads of a tile then
    // depend on the tile record only).
    std::vector<GTile> out(maxbin * NX, GTile{-1, 0, 0, -1});
    for (int x = 0; x < NX; ++x) {
        // ... and inside an XCD the clusters with the LONGEST tiles run first, whatever their total: the launch ends when the
        // last tile ends, so the tail should be made of the shortest tiles, not of a small cluster of long ones
        // (coarse buckets of 16 k-steps, stable: clusters of similar tile length keep the LPT order, which keeps the clusters
        //  of one group -- same operands -- close together)
        std::stable_sort(binc[x].begin(), binc[x].end(), [](const Cl& a, const Cl& b) { return a.len16 > b.len16; });
        size_t pos = (size_t)x;
        for (const Cl& c : binc[x])
            for (uint32_t t = c.begin; t < c.end; ++t, pos += NX) { GTile v = tiles[t]; v.pad = first_gemm[(size_t)v.group]; out[pos] = v; }
    }
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
    // Claim counters of this launch: a block of 16 ints of its own, ZERO when the kernel starts because the launch stream itself zeroed
    // it -- a ring per (device, stream), cleared one half at a time by a hipMemsetAsync on that stream when the launches enter the
    // half.  Stream order puts the memset behind every earlier launch that used the half and in front of every launch that will; no
    // launch depends on a previous one having left its counters in any state (a faulted or aborted kernel included), launches on other
    // streams or devices have rings of their own (VERDICT round 4 "weak" 9).  One 32 KB memset per 512 launches.
    int* ctr = nullptr;
    if (ntiles > (int32_t)grid) {                  // (only a claiming launch reads its counters)
        constexpr unsigned HALF = 512;
        struct Ring { int* base; unsigned next; };
        static std::mutex mu;
        static std::map<std::pair<int, hipStream_t>, Ring> rings;
        int dev = 0;
        DMRGX_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        Ring& r = rings[std::make_pair(dev, st)];
        if (!r.base) { DMRGX_HIP(hipMalloc((void**)&r.base, 2 * HALF * 16 * sizeof(int))); r.next = 0; }
        const unsigned slot = r.next % (2 * HALF);
        if (slot % HALF == 0) DMRGX_HIP(hipMemsetAsync(r.base + (size_t)slot * 16, 0, HALF * 16 * sizeof(int), st));
        r.next = slot + 1;
        ctr = r.base + (size_t)slot * 16;
    }
    constexpr size_t lds64 = (2 * 64 * (GG_BK + 2) + 2 * GG_BK * (64 + 16)) * sizeof(double) + 16, lds128 = (2 * 128 * (GG_BK + 2) + 2 * GG_BK * (128 + 16)) * sizeof(double) + 16;
    if (big) hipLaunchKernelGGL(ggemm_kernel_128, dim3(grid), dim3(512), lds128, st, d_tiles, d_groups, d_prods, ntiles, ctr GG_TRACE_ARG);
    else hipLaunchKernelGGL(ggemm_kernel_64, dim3(grid), dim3(256), lds64, st, d_tiles, d_groups, d_prods, ntiles, ctr GG_TRACE_ARG);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

}  // namespace dmrgx
