// Batched symmetric eigensolver for the reduced density matrices -- see symeig.h.  Three phases, every one of them batched over
// all matrices of a truncation step (the reference hands each block to LAPACK on rank 0, include/DMRGBlockContainer.hpp:1962-2003):
//
//  1. Householder tridiagonalisation  A = Q T Q^T,  ONE LAUNCH PER COLUMN (trid_step_kernel).  The eigenproblem of a DMRG step is a
//     latency chain, not a flop count (sum n^3 ~ 3e9 at m = 2048): a column step needs two dependent reductions over the whole
//     column, and the cheapest chip-wide barrier on this machine is a kernel boundary (~1.5 us; an in-kernel all-to-all exchange
//     costs ~3 us).  The classical two phases of a column (y = A v; rank-2 update A -= v w^T + w v^T) are skewed so that one
//     launch does both with one pass over the trailing matrix: every workgroup first finishes, redundantly, the O(n) vector work
//     of the column (w_{j-1} from y_{j-1}; the updated row j -> d_j, e_j, v_j, tau_j), then applies the update of reflector j-1
//     to its rows and multiplies the updated rows by v_j on the way (y_j).  n launches per matrix, all matrices in each launch.
//  2. Tridiagonal divide and conquer (Cuppen; the method of LAPACK's dstedc), level-synchronous: uniform-depth tree, leaves <= 32
//     solved by Jacobi in LDS, then per level: deflation (dc_deflate_kernel; the only sequential scan), secular roots by the
//     "middle way" iteration in shifted coordinates (one wave per root), Loewner weights and eigenvector columns, and ONE grouped
//     MFMA GEMM  Q_new = blockdiag(Q1, Q2) . U  in which deflated columns are unit columns of U (no gather / scatter kernels) and
//     the deflation rotations are applied to the ROWS of U.  Density matrices of DMRG states deflate almost completely.
//  3. Back-transformation  X = H_0 .. H_{n-3} Z  in blocks of 64 reflectors, compact-WY with T^-1 = striu(V^T V) + diag(1/tau):
//     T V^T is formed for all blocks at once, then two grouped GEMMs per block step.
//
// tools/proto_trid_dc.py is the numpy statement of exactly this data flow.
#include "symeig.h"
#include <chrono>
#include "ggemm.h"
#include <cfloat>
#include <cmath>
#include <numeric>

namespace dmrgx {
namespace {

constexpr int TRID_THREADS = 256, TRID_ROWS = 8;      // rows of the trailing matrix per workgroup (two per wave)
constexpr int TRID_PF = 8;                            // chunks of 64 columns requested per row before they are used
#ifndef DMRGX_DC_LEAF
#define DMRGX_DC_LEAF 16
#endif
constexpr int DC_LEAF = DMRGX_DC_LEAF;                // largest leaf of the divide-and-conquer tree.  Measured (same box; Rdms per step of configs[1] /
                                                      // truncation of cfg4real): 32: 2.00 / 8.85 ms, 16: 1.84 / 8.60, 8: 1.90 / 8.62 -- a Jacobi leaf of 16 costs a
                                                      // quarter of one of 32, which is more than the extra merge level takes
#ifndef DMRGX_DC_LEAF_THREADS
#define DMRGX_DC_LEAF_THREADS 128
#endif
constexpr int DC_LEAF_THREADS = DMRGX_DC_LEAF_THREADS;     // (leaf of 32: one wave measured 2x slower than 256 threads; leaf of 16: 128 threads as fast as 256)
#ifndef DMRGX_WY_NB
#define DMRGX_WY_NB 64
#endif
constexpr int WY_NB = DMRGX_WY_NB;                    // reflectors per block of the back-transformation.  -DDMRGX_WY_NB=128 (T assembled from two 64-halves, half as
                                                      // many dependent GEMM steps) measured SLOWER on the same box: 9.03-9.07 vs 8.72-8.74 ms per create at m = 2048
constexpr int WY_SUB = 64;                            // its T factor is inverted in halves: T = [T1, -T1 (V1^T V2) T2; 0, T2]
constexpr double DC_EPS = DBL_EPSILON;

struct TridMat {
    double *A, *VT, *y, *d, *e, *tau;                 // VT: row j = reflector v_j (zeros up to column j); y: 2 x n (double-buffered A v)
    int32_t n, lda, ldv, pad;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The same sum in every lane without the LDS crossbar (ds_bpermute costs ~100 cycles per step on the latency path of every column):
// lane bits 0, 1 by quad permutes, bits 2, 3 by row rotations (a rotation all-reduce: after +ror4 and +ror8 every lane of a row
// holds the row's sum), bits 4, 5 by the gfx950 row / half swaps.
#define DMRGX_DPP_ADD(a, CTRL)                                                                                                   \
    {                                                                                                                            \
        const int lo_ = __double2loint(a), hi_ = __double2hiint(a);                                                              \
        a += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi_, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo_, CTRL, 0xf, 0xf, false)); \
    }
__device__ __forceinline__ double wave_sum_fast(double a)
{
    DMRGX_DPP_ADD(a, 0xb1);      // quad_perm [1,0,3,2]
    DMRGX_DPP_ADD(a, 0x4e);      // quad_perm [2,3,0,1]
    DMRGX_DPP_ADD(a, 0x124);     // row_ror:4
    DMRGX_DPP_ADD(a, 0x128);     // row_ror:8
    int lo = __double2loint(a), hi = __double2hiint(a);
    auto l16 = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
    a = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
    lo = __double2loint(a); hi = __double2hiint(a);
    auto l32 = __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
    return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}
// one-barrier block sum: `red` must not be in use by waves that have not passed a barrier since they last read it
template <int NW>
__device__ __forceinline__ double block_sum1(double v, double* red, int tid)
{
    v = wave_sum_fast(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over the workgroup; `red` holds one slot per wave.  Two barriers: the first also publishes whatever the callers wrote to
// LDS before the call, the second makes the partial sums visible.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* red, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double* red, int tid)
{
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = fmax(s, red[w]);
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1. tridiagonalisation: launch j
// ---------------------------------------------------------------------------------------------------------------------------
// Stored state at the start of launch j: rows / columns >= j of A carry the updates of reflectors 0 .. j-2; y_{j-1} = A v_{j-1}
// (un-scaled) and v_{j-1}, tau_{j-1} are in memory.  Row j is read by everybody and written by nobody (it retires here), rows
// > j are each updated by exactly one wave, so the launch has no write-read race on A.
constexpr int TRID_MAXM = 32;                         // matrices per launch: their descriptors travel in the kernel argument segment
struct TridArgs { TridMat m[TRID_MAXM]; };            // (a descriptor table in memory costs every workgroup two dependent cold loads
                                                      //  before it can ask for its vectors: ~1.5 us of a ~5 us launch)
__global__ void __launch_bounds__(TRID_THREADS)
trid_step_kernel(const TridArgs args, int j)
{
    extern __shared__ double sh[];                  // w | v_{j-1} | row j, then v_j   (indexed by the absolute column)
    __shared__ double red[TRID_THREADS / 64];
    const TridMat& m = args.m[blockIdx.y];
    const int n = m.n, tid = threadIdx.x, g = blockIdx.x;
    // Workgroup g owns rows [8 g, 8 g + 8) of its matrix in EVERY launch (the grid does not shrink with j: workgroups whose rows have
    // all retired leave at once), so a row is always handled by the same position of the grid.  The last row block also does the
    // bookkeeping of the column (it is never retired before the matrix is done).
    const int glast = (n - 1) / TRID_ROWS;
    if (j >= n || g > glast || (TRID_ROWS * g + TRID_ROWS - 1 <= j && g != glast)) return;
    double* sw = sh;
    double* svp = sh + n;
    double* svj = sh + 2 * n;
    const double tau_p = j > 0 ? m.tau[j - 1] : 0.0;
    const double* yp = m.y + (size_t)((j + 1) & 1) * n;
    double* yc = m.y + (size_t)(j & 1) * n;
    const double* vprow = m.VT + (int64_t)(j > 0 ? j - 1 : 0) * m.ldv;
    const double* arow = m.A + (int64_t)j * m.lda;
    // the rows this wave will update.  Their first TRID_PF chunks of 64 columns are requested now, ahead of the column's vector work
    // (the loads depend on nothing but addresses; the update is latency-bound otherwise)
    const int lane = tid & 63, wave = tid >> 6;
    const int i0 = TRID_ROWS * g + 2 * wave;            // rows i0, i0+1, where they still belong to the trailing matrix (> j)
    const bool r0 = i0 > j && i0 < n, r1 = i0 + 1 > j && i0 + 1 < n;
    double* a0 = m.A + (int64_t)(r0 ? i0 : j) * m.lda;
    double* a1 = m.A + (int64_t)(r1 ? i0 + 1 : j) * m.lda;
    const int kb = j + 1 + lane;
    double pa0[TRID_PF], pa1[TRID_PF];
#pragma unroll
    for (int c = 0; c < TRID_PF; ++c) {
        const int k = kb + 64 * c;
        pa0[c] = (r0 && k < n) ? a0[k] : 0.0;
        pa1[c] = (r1 && k < n) ? a1[k] : 0.0;
    }
    double s = 0.0;
    for (int k = j + tid; k < n; k += TRID_THREADS) {
        const double vp = j > 0 ? vprow[k] : 0.0, y = j > 0 ? yp[k] : 0.0;
        svp[k] = vp; sw[k] = y; svj[k] = arow[k];
        s += y * vp;
    }
    s = block_sum<TRID_THREADS / 64>(s, red, tid);
    const double coef = 0.5 * tau_p * tau_p * s;     // w = tau y - (tau^2 (y.v) / 2) v
    for (int k = j + tid; k < n; k += TRID_THREADS) sw[k] = tau_p * sw[k] - coef * svp[k];
    __syncthreads();
    const double wj = sw[j], vpj = svp[j];
    double sig = 0.0;
    for (int k = j + tid; k < n; k += TRID_THREADS) {     // row j with the update of reflector j-1 applied
        const double r = svj[k] - vpj * sw[k] - wj * svp[k];
        svj[k] = r;
        if (k >= j + 2) sig += r * r;
    }
    sig = block_sum<TRID_THREADS / 64>(sig, red, tid);
    const double dj = svj[j];
    const double alpha = (j + 1 < n) ? svj[j + 1] : 0.0;
    double beta = alpha, tj = 0.0, scale = 0.0;          // dlarfg: H x = beta e_1, H = I - tau v v^T, v_1 = 1
    if (sig > 0.0) {
        const double nrm = sqrt(alpha * alpha + sig);
        beta = alpha >= 0.0 ? -nrm : nrm;
        tj = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    __syncthreads();                                    // everyone holds d_j and alpha before row j turns into v_j
    for (int k = j + tid; k < n; k += TRID_THREADS)     // a reflector with tau = 0 is stored as the zero vector (H = I)
        svj[k] = (k <= j || tj == 0.0) ? 0.0 : (k == j + 1 ? 1.0 : svj[k] * scale);
    __syncthreads();
    if (g == glast) {
        double* vt = m.VT + (int64_t)j * m.ldv;
        for (int k = tid; k < n; k += TRID_THREADS) vt[k] = k > j ? svj[k] : 0.0;
        if (tid == 0) { m.d[j] = dj; if (j + 1 < n) m.e[j] = beta; m.tau[j] = tj; }
    }
    // ---- own rows: A[i,:] -= v_{j-1}[i] w^T + w[i] v_{j-1}^T, then y_j[i] = A[i,:] . v_j -- two rows per wave
    if (!r0 && !r1) return;
    const double vp0 = r0 ? svp[i0] : 0.0, w0 = r0 ? sw[i0] : 0.0, vp1 = r1 ? svp[i0 + 1] : 0.0, w1 = r1 ? sw[i0 + 1] : 0.0;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int c = 0; c < TRID_PF; ++c) {
        const int k = kb + 64 * c;
        if (k < n) {
            const double wk = sw[k], vpk = svp[k], vjk = svj[k];
            const double x0 = pa0[c] - (vp0 * wk + w0 * vpk), x1 = pa1[c] - (vp1 * wk + w1 * vpk);
            if (r0) a0[k] = x0;
            if (r1) a1[k] = x1;
            acc0 += x0 * vjk; acc1 += x1 * vjk;
        }
    }
    for (int kc = kb + 64 * TRID_PF; kc < n + lane; kc += 64 * TRID_PF) {     // (kc - lane < n: wave-uniform trip count)
#pragma unroll
        for (int c = 0; c < TRID_PF; ++c) {
            const int k = kc + 64 * c;
            pa0[c] = (r0 && k < n) ? a0[k] : 0.0;
            pa1[c] = (r1 && k < n) ? a1[k] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < TRID_PF; ++c) {
            const int k = kc + 64 * c;
            if (k < n) {
                const double wk = sw[k], vpk = svp[k], vjk = svj[k];
                const double x0 = pa0[c] - (vp0 * wk + w0 * vpk), x1 = pa1[c] - (vp1 * wk + w1 * vpk);
                if (r0) a0[k] = x0;
                if (r1) a1[k] = x1;
                acc0 += x0 * vjk; acc1 += x1 * vjk;
            }
        }
    }
    acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
    if (lane == 0) { if (r0) yc[i0] = acc0; if (r1) yc[i0 + 1] = acc1; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1b. tridiagonalisation with the matrices RESIDENT IN LDS: one persistent launch
// ---------------------------------------------------------------------------------------------------------------------------
// The launch-per-column kernel above pays, per column, a kernel boundary plus a chain of cold loads (~4 us) and streams the whole
// trailing matrix through the fabric twice (read + write-back: ~5 us on average at m = 2048).  Here the rows of a matrix are dealt
// cyclically over G workgroups (one per CU, up to 160 KB of LDS each: 16 rows of a 1037 x 1037 matrix) and never leave the chip;
// per column the workgroups exchange only what the vector work needs -- y_j = A v_j (each publishes the entries of its rows) and
// the next pivot row (published by its owner while it updates it) -- as 8-byte {epoch, half of a double} granules written with
// agent-scope atomic stores and polled with agent-scope atomic loads: the data is the flag, no fence, no ordering assumption
// (cdna_hip_programming.md section 6, Guideline 16, form R2).  Buffers are double-buffered by column parity: a workgroup can be at
// most one column ahead of the slowest one, because publishing column j+1 needs everybody's column j.  Every spin is bounded: a
// workgroup that is not served in time (its partners are not resident -- e.g. another process holds CUs with a persistent kernel of
// its own) sets the status word and leaves; the host then repeats the step with the launch-per-column kernel (A is only read here).
typedef unsigned long long u64;
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef u64 __attribute__((address_space(1))) gu64;
#define DMRGX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int TC_THREADS = 512, TC_WAVES = TC_THREADS / 64;
constexpr int TC_SPIN_LIMIT = 1 << 19;               // polls of one batch of granules before giving up (~0.5 s)
#ifndef TC_FIRST_SLEEP
#define TC_FIRST_SLEEP 10                            // s_sleep units (64 clocks) before the first poll of a column: the partners' publications are still on their way
#endif                                               // then, and polls that come back empty are traffic in their way: 0 / 6 / 8 / 10 / 12 / 14 -> 752 / 721 / 709 / 701 / 706 / 715 us
#ifndef TC_RETRY_SLEEP                               // per call at m = 512, 4556 / 4309 / 4265 / 4262 / 4291 / 4335 at m = 2048 (profiles/r05_trid_first_poll_delay.txt)
#define TC_RETRY_SLEEP 1                             // between polls of what is still missing (0 / 1 / 2 / 4: 700 / 701 / 706 / 765 us)
#endif
constexpr int TC_KB = 4;                             // columns per thread whose granules are in flight together
constexpr int TC_MB = 4;                             // chunks of 64 columns of the row pass whose LDS reads are in flight together
constexpr int TC_VECS = 4;                           // O(n) vectors every workgroup keeps beside its rows

struct TcMat {
    const double* A; double *VT, *d, *e, *tau;
    u64 *ybuf, *rowbuf;                               // 2 (parity) x n x 2 granules each
    int32_t n, lda, ldv, G, wg0, pad;
};
struct TcArgs { TcMat m[TRID_MAXM]; int32_t* status; long long* prof; int32_t nm, fault; };      // prof: developer aid (phase clocks of one workgroup), else null;
                                                                                                // fault: test hook, one workgroup withholds one publication

__device__ __forceinline__ void put_f64(u64* g, unsigned epoch, double v)
{
    const u64 b = (u64)__double_as_longlong(v), e = (u64)epoch << 32;
    __hip_atomic_store((gu64*)g, e | (b & 0xffffffffull), DMRGX_RLX_AGENT);
    __hip_atomic_store((gu64*)(g + 1), e | (b >> 32), DMRGX_RLX_AGENT);
}
__device__ __forceinline__ bool wait_f64(const u64* g, unsigned epoch, double& v)
{
    for (int spins = 0; spins < TC_SPIN_LIMIT; ++spins) {
        const u64 x0 = __hip_atomic_load((gu64*)g, DMRGX_RLX_AGENT), x1 = __hip_atomic_load((gu64*)(g + 1), DMRGX_RLX_AGENT);
        if ((unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch) { v = __longlong_as_double((long long)((x0 & 0xffffffffull) | (x1 << 32))); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    v = 0.0;
    return false;
}

// (Measured and dropped: the rows in REGISTERS instead of LDS -- 256-thread workgroups, one wave per SIMD, 6 rows x 17 chunks per
// lane.  The f64 update of a lane's ~100 elements is then one dependent-issue stream per SIMD: 2.5 us per column against ~1 us for
// eight waves over LDS-resident rows, and the whole column 9.9 us against 4.9.)
__global__ void __launch_bounds__(TC_THREADS)
trid_coop_kernel(const TcArgs args)
{
    extern __shared__ __attribute__((aligned(16))) double shc[];      // y / w (two buffers) | v (two buffers) | the rows of this workgroup
    __shared__ double red_s[TC_WAVES], red_g[TC_WAVES], piv[2];
    __shared__ int sfail;
    const int b = blockIdx.x;
    int mi = 0;
    for (; mi < args.nm; ++mi) if (b >= args.m[mi].wg0 && b < args.m[mi].wg0 + args.m[mi].G) break;
    if (mi == args.nm) return;
    const TcMat& m = args.m[mi];
    const int n = m.n, G = m.G, g = b - m.wg0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = (n + 1) & ~1;                        // even stride: the row pass moves two columns per lane (ds_read_b128 / ds_write_b128)
    double* sw = shc;
    double* sw_next = shc + nv;
    double* svp = shc + 2 * (size_t)nv;
    double* svj = shc + 3 * (size_t)nv;
    double* rows = shc + TC_VECS * (size_t)nv;          // row t of this workgroup = row g + G t of the matrix
    const int cnt = g < n ? (n - g + G - 1) / G : 0;
    // Who runs how long.  A workgroup publishes for as long as it owns a row below the pivot: rows i > j contribute y_j[i], and row j + 1
    // is published by its owner.  The owner of row n - 1 (`glast`) therefore publishes in every column, every other workgroup needs its
    // granule of every column, and it can never fall two columns behind unnoticed -- so it is the one that writes d, e, tau and V^T; a
    // workgroup whose last row has become the pivot (or that has no row at all) LEAVES: it would only consume, nobody would wait for it,
    // and a lagging pure consumer finds its double-buffered slots overwritten by epochs j + 2 / j + 3 and spins to the time-out
    // (round 3 had workgroup 0 do the bookkeeping to the end).
    const int glast = (n - 1) % G, ilast = cnt > 0 ? g + G * (cnt - 1) : -1;
    if (cnt == 0) return;
    for (int t = 0; t < cnt; ++t) {
        const double* src = m.A + (int64_t)(g + G * t) * m.lda;
        for (int k = tid; k < nv; k += TC_THREADS) rows[(size_t)t * nv + k] = k < n ? src[k] : 0.0;
    }
    for (int k = tid; k < nv; k += TC_THREADS) { svj[k] = 0.0; svp[k] = 0.0; sw[k] = 0.0; sw_next[k] = 0.0; }
    if (g == 0) for (int k = tid; k < n; k += TC_THREADS) put_f64(m.rowbuf + 2 * (size_t)k, 1u, m.A[k]);     // row 0, epoch 1
    if (tid == 0) sfail = 0;
    __syncthreads();
    double tau_p = 0.0, out_d = 0.0, out_e = 0.0;
    const bool prof = args.prof && mi == 0 && g == (G > 1 ? 1 : 0) && tid == 0;
    long long pt[4] = {0, 0, 0, 0}, pc = 0;
    for (int j = 0; j < n; ++j) {
        if (g != glast && j >= ilast) {                                // (uniform per workgroup) nothing left to publish: see above
            if (prof) for (int q = 0; q < 4; ++q) args.prof[q] = pt[q] / (j > 0 ? j : 1);      // (developer aid: mean ticks per column of the columns it took part in)
            return;
        }
        if (prof) pc = wall_clock64();
        { double* t = svp; svp = svj; svj = t; }                      // svp = v_{j-1}
        // `glast` (which everybody waits for in every column) writes column j-1's results HERE, where it would otherwise only wait for
        // the partners' publications, instead of in front of its own rows
        if (g == glast && j > 0) {
            double* vt = m.VT + (int64_t)(j - 1) * m.ldv;
            for (int k = tid; k < n; k += TC_THREADS) vt[k] = k > j - 1 ? svp[k] : 0.0;
            if (tid == 0) { m.d[j - 1] = out_d; m.e[j - 1] = out_e; m.tau[j - 1] = tau_p; }
        }
        const u64* rb = m.rowbuf + (size_t)(j & 1) * 2 * n;              // row j, epoch j+1
        const u64* yb = m.ybuf + (size_t)((j + 1) & 1) * 2 * n;          // y_{j-1}, epoch j
        double s = 0.0;
        bool ok = true, lap = false;
        // every granule this thread needs is requested before the first tag is looked at (TC_KB columns x 4 loads in flight), and only
        // the columns that were not there yet are asked for again
#if TC_FIRST_SLEEP > 0
        if (j > 0) __builtin_amdgcn_s_sleep(TC_FIRST_SLEEP);      // the partners' publications are in flight: a poll that leaves now comes back empty and is in their way
#endif
        for (int k0 = j + tid; k0 < n; k0 += TC_THREADS * TC_KB) {
            unsigned pending = 0;
#pragma unroll
            for (int c = 0; c < TC_KB; ++c) if (k0 + c * TC_THREADS < n) pending |= 1u << c;
            for (int spins = 0; pending; ++spins) {
                u64 x[TC_KB][4];
#pragma unroll
                for (int c = 0; c < TC_KB; ++c) {
                    if (!(pending >> c & 1)) continue;
                    const size_t k = (size_t)(k0 + c * TC_THREADS);
                    x[c][0] = __hip_atomic_load((gu64*)(rb + 2 * k), DMRGX_RLX_AGENT);
                    x[c][1] = __hip_atomic_load((gu64*)(rb + 2 * k + 1), DMRGX_RLX_AGENT);
                    if (j > 0) { x[c][2] = __hip_atomic_load((gu64*)(yb + 2 * k), DMRGX_RLX_AGENT); x[c][3] = __hip_atomic_load((gu64*)(yb + 2 * k + 1), DMRGX_RLX_AGENT); }
                }
#pragma unroll
                for (int c = 0; c < TC_KB; ++c) {
                    if (!(pending >> c & 1)) continue;
                    const unsigned er = (unsigned)(j + 1), ey = (unsigned)j;
                    bool ready = (unsigned)(x[c][0] >> 32) == er && (unsigned)(x[c][1] >> 32) == er;
                    // an epoch NEWER than the one waited for is not "not yet": the slot was overwritten, this workgroup has been lapped
                    // -- a protocol error, reported at once instead of after 0.5 s of spinning
                    bool lapped = (unsigned)(x[c][0] >> 32) > er || (unsigned)(x[c][1] >> 32) > er;
                    if (j > 0) {
                        ready = ready && (unsigned)(x[c][2] >> 32) == ey && (unsigned)(x[c][3] >> 32) == ey;
                        lapped = lapped || (unsigned)(x[c][2] >> 32) > ey || (unsigned)(x[c][3] >> 32) > ey;
                    }
                    if (lapped) { ok = false; lap = true; pending = 0; break; }
                    if (!ready) continue;
                    const int k = k0 + c * TC_THREADS;
                    const double a = __longlong_as_double((long long)((x[c][0] & 0xffffffffull) | (x[c][1] << 32)));
                    const double y = j > 0 ? __longlong_as_double((long long)((x[c][2] & 0xffffffffull) | (x[c][3] << 32))) : 0.0;
                    sw[k] = y; svj[k] = a;
                    pending &= ~(1u << c);
                }
                if (pending) {
                    if (spins > TC_SPIN_LIMIT) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(TC_RETRY_SLEEP);
                }
            }
        }
        // INVARIANT: the vector arithmetic of a column is done redundantly by every workgroup of the matrix and must be BIT-IDENTICAL in
        // all of them (same thread -> column map, same summation order): they never exchange v_j or w, so a one-ulp disagreement is
        // never repaired and the recurrence amplifies it (found the hard way: adding y.v in granule-arrival order broke Tr T = Tr A).
        for (int k = j + tid; k < n; k += TC_THREADS) s += sw[k] * svp[k];
        if (prof) { const long long c = wall_clock64(); pt[0] += c - pc; pc = c; }
        if (!ok) sfail = lap ? 2 : 1;
        s = block_sum1<TC_WAVES>(s, red_s, tid);          // (its barrier also publishes the y and the pivot row the other threads just stored)
        if (sfail) { if (tid == 0) __hip_atomic_store(args.status, sfail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }     // 1: timed out, 2: lapped
        const double coef = 0.5 * tau_p * tau_p * s;
        const double vpj = svp[j], wj = tau_p * sw[j] - coef * vpj;     // w_{j-1}[j], straight from y[j]: no barrier between w and the row update
        double sig = 0.0;
        for (int k = j + tid; k < n; k += TC_THREADS) {
            const double vpk = svp[k], wk = tau_p * sw[k] - coef * vpk;
            const double r = svj[k] - vpj * wk - wj * vpk;          // row j with the update of reflector j-1 applied
            sw_next[k] = wk;
            svj[k] = r;
            if (k >= j + 2) sig += r * r;
            if (k == j) piv[0] = r;
            if (k == j + 1) piv[1] = r;
        }
        if (j + 1 >= n && tid == 0) piv[1] = 0.0;
        sig = block_sum1<TC_WAVES>(sig, red_g, tid);
        const double dj = piv[0], alpha = piv[1];
        double beta = alpha, tj = 0.0, scale = 0.0;
        if (sig > 0.0) {
            const double nrm = sqrt(alpha * alpha + sig);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tj = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int k = j + tid; k < n; k += TC_THREADS) svj[k] = (k <= j || tj == 0.0) ? 0.0 : (k == j + 1 ? 1.0 : svj[k] * scale);     // own columns only
        __syncthreads();
        { double* t = sw; sw = sw_next; sw_next = t; }          // sw = w_{j-1} (the y buffer of this column is the w buffer of the next)
        out_d = dj; out_e = beta;                                // (written out by `glast` at the top of the next column, see there)
        tau_p = tj;
        if (prof) { const long long c = wall_clock64(); pt[1] += c - pc; pc = c; }
        // ---- own rows (in LDS): update with reflector j-1, multiply by v_j; the owner of row j+1 publishes it while it is in hand.
        //      The rows still alive (t >= tmin) are dealt over the waves afresh every column -- wave w takes rows tmin + w and
        //      tmin + w + 8 -- so that retired rows do not leave waves idle; the two rows of a wave share the reads of w, v_{j-1}, v_j.
        u64* rbn = m.rowbuf + (size_t)((j + 1) & 1) * 2 * n;
        u64* ybn = m.ybuf + (size_t)(j & 1) * 2 * n;
        const int tmin = j >= g ? (j - g) / G + 1 : 0;                 // first row of this workgroup below the pivot (row j+1, if it is ours, is this one)
        const int ks = (j + 1) & ~1;                                   // first (even) column of the pass; column j, if included, is dead
        for (int t0 = tmin + wave; t0 < cnt; t0 += 2 * TC_WAVES) {
            const int t1 = t0 + TC_WAVES;
            const int i0 = g + G * t0, i1 = g + G * t1;
            const bool a1 = t1 < cnt;
            double* row0 = rows + (size_t)t0 * nv;
            double* row1 = rows + (size_t)(a1 ? t1 : t0) * nv;
            const double vp0 = svp[i0], w0 = sw[i0], vp1 = a1 ? svp[i1] : 0.0, w1 = a1 ? sw[i1] : 0.0;
            const bool pub0 = i0 == j + 1;
            double acc0 = 0.0, acc1 = 0.0;
            for (int kc = ks + 2 * lane; kc - 2 * lane < n; kc += 128 * TC_MB) {       // (wave-uniform trip count)
                dbl2 wk[TC_MB], vpk[TC_MB], vjk[TC_MB], x0[TC_MB], x1[TC_MB];
#pragma unroll
                for (int c = 0; c < TC_MB; ++c) {
                    const int k = min(kc + 128 * c, nv - 2);
                    wk[c] = *(const dbl2*)(sw + k); vpk[c] = *(const dbl2*)(svp + k); vjk[c] = *(const dbl2*)(svj + k);
                    x0[c] = *(const dbl2*)(row0 + k); x1[c] = *(const dbl2*)(row1 + k);
                }
#pragma unroll
                for (int c = 0; c < TC_MB; ++c) {
                    const int k = kc + 128 * c;
                    if (k < n) {                       // (the pad column of an odd n holds zeros in every vector and row: it stays zero)
                        const dbl2 y0 = x0[c] - (vp0 * wk[c] + w0 * vpk[c]), y1 = x1[c] - (vp1 * wk[c] + w1 * vpk[c]);
                        *(dbl2*)(row0 + k) = y0;
                        acc0 += y0.x * vjk[c].x + y0.y * vjk[c].y;          // (v_j is zero in column j)
                        if (pub0) { if (k > j) put_f64(rbn + 2 * (size_t)k, (unsigned)(j + 2), y0.x); if (k + 1 < n) put_f64(rbn + 2 * (size_t)(k + 1), (unsigned)(j + 2), y0.y); }
                        if (a1) { *(dbl2*)(row1 + k) = y1; acc1 += y1.x * vjk[c].x + y1.y * vjk[c].y; }
                    }
                }
            }
            acc0 = wave_sum_fast(acc0); acc1 = wave_sum_fast(acc1);
            const bool withhold = args.fault && mi == 0 && j == 3 && i0 == n - 1;      // (test hook: the last row's y_3 never arrives -> every partner times out)
            if (lane == 0 && !withhold) { put_f64(ybn + 2 * (size_t)i0, (unsigned)(j + 1), acc0); if (a1) put_f64(ybn + 2 * (size_t)i1, (unsigned)(j + 1), acc1); }
        }
        if (prof) { const long long c = wall_clock64(); pt[2] += c - pc; pc = c; }
        __syncthreads();
        if (prof) { const long long c = wall_clock64(); pt[3] += c - pc; pc = c; }
    }
    if (g == glast && n > 0) {                                     // the last column's results (its reflector is the zero vector)
        double* vt = m.VT + (int64_t)(n - 1) * m.ldv;
        for (int k = tid; k < n; k += TC_THREADS) vt[k] = 0.0;
        if (tid == 0) { m.d[n - 1] = out_d; m.tau[n - 1] = tau_p; }
    }
    if (prof) for (int q = 0; q < 4; ++q) args.prof[q] = pt[q] / (n > 0 ? n : 1);
}

// dst (n x n, ld) = src^T for every matrix (V from V^T)
struct SqPair { const double* src; double* dst; int32_t n, lds, ldd, pad; };
__global__ void __launch_bounds__(256) transpose_sq_kernel(const SqPair* __restrict__ prs)
{
    __shared__ double t[32][33];
    const SqPair p = prs[blockIdx.z];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= p.n || c0 >= p.n) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) t[r][tx] = (r0 + r < p.n && c0 + tx < p.n) ? p.src[(int64_t)(r0 + r) * p.lds + c0 + tx] : 0.0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) if (c0 + r < p.n && r0 + tx < p.n) p.dst[(int64_t)(c0 + r) * p.ldd + r0 + tx] = t[tx][r];
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3. compact-WY factor of a block of reflectors:  Tneg = -(striu(G) + diag(1 / tau))^-1,  G = V_b^T V_b  (64 x 64, row-major)
// ---------------------------------------------------------------------------------------------------------------------------
struct WyBlock { const double* G; const double* tau; double* Tneg; int32_t kb, ld, zero_below, pad; };      // one 64 x 64 diagonal sub-block; G and Tneg with row stride ld;
                                                                                                      // zero_below: rows of the (structurally zero) block under it to clear
__global__ void __launch_bounds__(WY_SUB) wy_tinv_kernel(const WyBlock* __restrict__ blocks)
{
    __shared__ double S[WY_SUB][WY_SUB + 1];          // upper triangle + diagonal: S; strictly lower triangle: T^T (T[r][c] at [c][r])
    const WyBlock b = blocks[blockIdx.x];
    const int c = threadIdx.x, kb = b.kb;
    for (int r = 0; r < WY_SUB; ++r) {
        double v = 0.0;
        if (r < kb && c < kb) {
            if (c > r) v = b.G[(int64_t)r * b.ld + c];
            else if (c == r) { const double t = b.tau[r]; v = t != 0.0 ? 1.0 / t : 1.0; }
        } else if (r == c) v = 1.0;
        if (c >= r) S[r][c] = v;
    }
    __syncthreads();
    // column c of S^-1 by back substitution (S upper triangular): t_c = 1 / S_cc;  t_r = -(sum_{l = r+1 .. c} S_rl t_l) / S_rr.
    // Thread c writes row c of the lower triangle only and reads the upper triangle, which nobody writes: no barrier in the loop.
    const double tcc = 1.0 / S[c][c];
    for (int r = WY_SUB - 2; r >= 0; --r) {
        if (r < c) {
            double acc = S[r][c] * tcc;
            for (int l = r + 1; l < c; ++l) acc += S[r][l] * S[c][l];
            S[c][r] = -acc / S[r][r];
        }
    }
    for (int r = 0; r < kb; ++r) {
        if (c >= kb) break;
        const double t = r < c ? S[c][r] : (r == c ? tcc : 0.0);
        b.Tneg[(int64_t)r * b.ld + c] = -t;
    }
    for (int r = 0; r < b.zero_below; ++r) b.Tneg[(int64_t)(WY_SUB + r) * b.ld + c] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2. divide and conquer
// ---------------------------------------------------------------------------------------------------------------------------
struct DcMat {
    double* Q[2];                  // eigenvector buffers (block diagonal while the tree is climbed); level l lives in Q[l & 1]
    double* U;                     // the merge matrices of one level (block diagonal)
    const double *d, *e;           // the tridiagonal matrix (unscaled)
    double *dcur, *scale, *w;      // eigenvalues of the current nodes (scaled); the scale factor; output spectrum
    double *dl, *zl, *tau, *lam, *zhat, *cnorm, *dval, *rc, *rs, *mrho;     // per position of the matrix, length n
    int32_t *org, *pcol, *dcol, *rowpole, *colroot, *pcolmap, *rcp, *rcj, *mk, *mrot;
    int32_t n, ldq[2], ldu;
};
struct DcLeaf { int32_t mat, lo, hi, buf; };
struct DcMerge { int32_t mat, lo, mid, hi, src, pad; };

__global__ void __launch_bounds__(256) dc_scale_kernel(const DcMat* __restrict__ mats)
{
    __shared__ double red[4];
    const DcMat m = mats[blockIdx.x];
    double v = 0.0;
    for (int i = threadIdx.x; i < m.n; i += 256) { v = fmax(v, fabs(m.d[i])); if (i + 1 < m.n) v = fmax(v, fabs(m.e[i])); }
    v = block_max<4>(v, red, threadIdx.x);
    if (threadIdx.x == 0) m.scale[0] = v > 0.0 ? v : 1.0;
}

// Leaf: T[lo:hi] with the rank-one couplings to its neighbours taken off the end diagonals (T = diag(T1', T2') + |beta| u u^T,
// u = e_last(1) + sign(beta) e_first(2)), diagonalised by cyclic Jacobi on the dense 32 x 32 array in LDS.
__global__ void __launch_bounds__(DC_LEAF_THREADS) dc_leaf_kernel(const DcMat* __restrict__ mats, const DcLeaf* __restrict__ leaves)
{
    constexpr int P = DC_LEAF, H = P / 2;
    __shared__ double S[P][P + 1], R[P][P + 1];
    __shared__ double rc[H], rs[H];
    __shared__ double red[4];
    __shared__ int rnk[P];
    const DcLeaf lf = leaves[blockIdx.x];
    const DcMat m = mats[lf.mat];
    const int p = lf.hi - lf.lo, tid = threadIdx.x;
    const double inv = 1.0 / m.scale[0];
    for (int e = tid; e < P * P; e += DC_LEAF_THREADS) { const int i = e / P, c = e % P; S[i][c] = 0.0; R[i][c] = i == c ? 1.0 : 0.0; }
    __syncthreads();
    if (tid < p) {
        const int gi = lf.lo + tid;
        double dv = m.d[gi] * inv;
        if (tid == 0 && lf.lo > 0) dv -= fabs(m.e[lf.lo - 1] * inv);
        if (tid == p - 1 && lf.hi < m.n) dv -= fabs(m.e[lf.hi - 1] * inv);
        S[tid][tid] = dv;
        if (tid + 1 < p) { const double ev = m.e[gi] * inv; S[tid][tid + 1] = ev; S[tid + 1][tid] = ev; }
    }
    __syncthreads();
    // round-robin tournament over pe = p rounded up to even players (a leaf of 17 needs 17 rounds of 9 pairs per sweep, not the 31 x 16
    // of the full 32 x 32 array; the odd player out meets the padding index p, whose row and column are zero: no rotation)
    const int pe = (p + 1) & ~1, hp = pe / 2, pm = pe - 1;
    // (pair, index) of the elements this thread rotates in every round: fixed for the whole solve -- the divisions by the run-time pe
    // and the modulo of the pairing were a third of a round's instructions
    constexpr int NE = (H * P + DC_LEAF_THREADS - 1) / DC_LEAF_THREADS;
    int et[NE], ei[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) { const int e = tid + u * DC_LEAF_THREADS; et[u] = e < hp * pe ? e / pe : -1; ei[u] = e % pe; }
    for (int sweep = 0; sweep < 20; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int e = tid; e < P * P; e += DC_LEAF_THREADS) { const int i = e / P, c = e % P; const double v = S[i][c]; if (i == c) dg += v * v; else off += v * v; }
        off = block_sum<DC_LEAF_THREADS / 64>(off, red, tid);
        dg = block_sum<DC_LEAF_THREADS / 64>(dg, red, tid);
        if (off <= 2e-31 * dg || off == 0.0) break;          // off-diagonal norm <= 2 eps |T|; uniform over the workgroup
        for (int r = 0; r < pe - 1; ++r) {
            auto pair_of = [&](int t, int& a, int& bq) {          // 0 <= r < pm, 1 <= t < hp: (r + t) mod pm and (r - t) mod pm by one conditional step
                if (t == 0) { a = pm; bq = r; } else { a = r + t; if (a >= pm) a -= pm; bq = r - t; if (bq < 0) bq += pm; }
                if (a > bq) { const int x = a; a = bq; bq = x; }
            };
            if (tid < hp) {                                  // pair tid of round r
                int a, bq;
                pair_of(tid, a, bq);
                const double apq = S[a][bq], app = S[a][a], aqq = S[bq][bq];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double th = 0.5 * (aqq - app) / apq;
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0); s = t * c;
                }
                rc[tid] = c; rs[tid] = s;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NE; ++u) {                                 // columns: S <- S J, R <- R J
                const int t = et[u], i = ei[u];
                if (t < 0) continue;
                int a, bq;
                pair_of(t, a, bq);
                const double c = rc[t], s = rs[t];
                const double sp = S[i][a], sq = S[i][bq];
                S[i][a] = c * sp - s * sq; S[i][bq] = s * sp + c * sq;
                const double rp = R[i][a], rq = R[i][bq];
                R[i][a] = c * rp - s * rq; R[i][bq] = s * rp + c * rq;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NE; ++u) {                                 // rows: S <- J^T S
                const int t = et[u], i = ei[u];
                if (t < 0) continue;
                int a, bq;
                pair_of(t, a, bq);
                const double c = rc[t], s = rs[t];
                const double sp = S[a][i], sq = S[bq][i];
                S[a][i] = c * sp - s * sq; S[bq][i] = s * sp + c * sq;
            }
            __syncthreads();
        }
    }
    // ascending order; the padding indices >= p never mixed with the real ones (their rows and columns are exactly zero)
    if (tid < p) {
        const double v = S[tid][tid];
        int r = 0;
        for (int q = 0; q < p; ++q) { const double u = S[q][q]; r += (u < v) || (u == v && q < tid); }
        rnk[tid] = r;
        m.dcur[lf.lo + r] = v;
    }
    __syncthreads();
    double* Q = m.Q[lf.buf];
    const int ldq = m.ldq[lf.buf];
    for (int e = tid; e < p * p; e += DC_LEAF_THREADS) { const int i = e / p, c = e % p; Q[(int64_t)(lf.lo + i) * ldq + lf.lo + rnk[c]] = R[i][c]; }
}

// (A one-wave implicit-QL leaf -- no barrier at all, the rotation recurrence redundantly in every lane -- was measured in round 3 and ran
//  the same 250 us per call as the Jacobi leaf above; it is not part of the library.)

// Deflation of one merge (LAPACK dlaed2 in this solver's data flow): z from the children's boundary rows, poles ranked by brute-
// force counting (no sortedness assumed), type-1 (rho |z_i| tiny) and type-2 (two close poles: one Givens rotation moves the
// weight to one of them) deflation in one sequential scan by thread 0 -- the only serial part of the solver.
__global__ void __launch_bounds__(1024) dc_deflate_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges, int nlmax)
{
    extern __shared__ double sh[];
    double* D = sh;
    double* z = sh + nlmax;
    double* ds = sh + 2 * (size_t)nlmax;
    double* zs = sh + 3 * (size_t)nlmax;
    int* ord = (int*)(sh + 4 * (size_t)nlmax);
    __shared__ double red[32];
    __shared__ int s_unsorted;
    const DcMerge mg = merges[blockIdx.x];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, n1 = mg.mid - mg.lo, nl = mg.hi - mg.lo, tid = threadIdx.x;
    const double beta = m.e[mg.mid - 1] / m.scale[0];
    const double rho = 2.0 * fabs(beta), sgn = beta < 0.0 ? -1.0 : 1.0;
    const double* Qs = m.Q[mg.src];
    const int ldq = m.ldq[mg.src];
    const double* row1 = Qs + (int64_t)(mg.mid - 1) * ldq + lo;      // last row of child 1 (columns lo .. mid-1)
    const double* row2 = Qs + (int64_t)mg.mid * ldq + lo;            // first row of child 2 (columns mid .. hi-1)
    double dmax = 0.0, zmax = 0.0;
    for (int i = tid; i < nl; i += 1024) {
        const double dv = m.dcur[lo + i];
        const double zv = (i < n1 ? row1[i] : sgn * row2[i]) * 0.70710678118654752440;
        D[i] = dv; z[i] = zv;
        dmax = fmax(dmax, fabs(dv)); zmax = fmax(zmax, fabs(zv));
        m.rowpole[lo + i] = -1;
    }
    {   // both maxima behind one pair of barriers
        dmax = wave_max(dmax); zmax = wave_max(zmax);
        if (tid == 0) s_unsorted = 0;
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = dmax; red[16 + (tid >> 6)] = zmax; }
        __syncthreads();
        dmax = red[0]; zmax = red[16];
#pragma unroll
        for (int w = 1; w < 16; ++w) { dmax = fmax(dmax, red[w]); zmax = fmax(zmax, red[16 + w]); }
    }
    // Rank of every pole in the union (ties by index).  The children's spectra are sorted ascending -- dc_leaf / dc_norm_rank write them
    // by rank -- so an entry's rank is its own position plus the number of the OTHER child's entries in front of it: a binary search
    // instead of nl comparisons (13 us of a launch at nl = 300, 160 at nl = 1062).  Verified, not assumed: a child that is not
    // sorted sends the merge through the brute-force count.
    for (int i = tid; i + 1 < nl; i += 1024) if (i + 1 != n1 && D[i] > D[i + 1]) s_unsorted = 1;
    __syncthreads();
    if (!s_unsorted) {
        for (int i = tid; i < nl; i += 1024) {
            const double di = D[i];
            int a, b, r;
            if (i < n1) { a = n1; b = nl; while (a < b) { const int c = (a + b) >> 1; if (D[c] < di) a = c + 1; else b = c; } r = i + (a - n1); }       // ties: the other child's index is larger
            else { a = 0; b = n1; while (a < b) { const int c = (a + b) >> 1; if (D[c] <= di) a = c + 1; else b = c; } r = (i - n1) + a; }             // ties: ... is smaller
            ord[r] = i;
        }
    } else {
        for (int i = tid; i < nl; i += 1024) {
            const double di = D[i];
            int r = 0;
            for (int q = 0; q < nl; ++q) { const double dq = D[q]; r += (dq < di) || (dq == di && q < i); }
            ord[r] = i;
        }
    }
    __syncthreads();
    for (int s = tid; s < nl; s += 1024) { const int c = ord[s]; ds[s] = D[c]; zs[s] = z[c]; }
    __syncthreads();
    const double tol = 8.0 * DC_EPS * fmax(dmax, zmax);
    // ---- fast path: no pair of neighbouring poles needs a type-2 rotation (the usual case without exact degeneracies), so the
    //      deflation scan has no sequential dependence: type-1 flags, a prefix count, and every pole / deflated entry written by
    //      its own thread.  One neighbour pair that would rotate sends the whole merge through the sequential scan below.
    int* pre = ord + nlmax;                               // exclusive count of non-deflated entries in front of sorted position s
    int* nf = pre + nlmax;                                // sorted positions of the non-deflated entries, in order
    __shared__ int wtot[16], s_any2, s_k;
    const bool all_defl = rho * zmax <= tol;
    {
        const int per = (nl + 1023) / 1024, s0 = tid * per;
        int c = 0;
        for (int u = 0; u < per; ++u) { const int sp = s0 + u; if (sp < nl && !(all_defl || rho * fabs(zs[sp]) <= tol)) ++c; }
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += v; }
        if ((tid & 63) == 63) wtot[tid >> 6] = inc;
        if (tid == 0) s_any2 = 0;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wtot[w];
        int run = base + inc - c;
        for (int u = 0; u < per; ++u) {
            const int sp = s0 + u;
            if (sp >= nl) break;
            pre[sp] = run;
            if (!(all_defl || rho * fabs(zs[sp]) <= tol)) { nf[run] = sp; ++run; }
        }
        if (tid == 1023) s_k = run;
        __syncthreads();
    }
    const int kfast = s_k;
    for (int p = 1 + tid; p < kfast; p += 1024) {         // would the scan rotate the pair (nf[p-1], nf[p])?
        const int a = nf[p - 1], b = nf[p];
        double s_ = zs[a], c_ = zs[b];
        const double tau = hypot(c_, s_), t = ds[b] - ds[a];
        c_ /= tau; s_ = -s_ / tau;
        if (fabs(t * c_ * s_) <= tol) s_any2 = 1;
    }
    __syncthreads();
    if (!s_any2) {
        for (int sp = tid; sp < nl; sp += 1024) {
            const bool d1 = all_defl || rho * fabs(zs[sp]) <= tol;
            const int c = ord[sp];
            if (d1) { const int t = sp - pre[sp]; m.dval[lo + t] = ds[sp]; m.dcol[lo + t] = c; }
            else { const int kk = pre[sp]; m.dl[lo + kk] = ds[sp]; m.zl[lo + kk] = zs[sp]; m.pcol[lo + kk] = c; m.rowpole[lo + c] = kk; }
        }
        if (tid == 0) { m.mk[lo] = kfast; m.mrot[lo] = 0; m.mrho[lo] = rho; }
        return;
    }
    // ---- sequential scan (thread 0), everything it touches in LDS and the pole it carries in registers: the test
    //      |t c s| <= tol with c = z_jj / tau, s = z_pj / tau is done as |t z_jj z_pj| <= tol tau^2 (no square root, no division
    //      unless the pair really rotates); the lists go to memory in parallel afterwards.
    int* polepos = pre;                                  // (the fast path's arrays are free again)
    int* deflpos = nf;
    int* rcp_s = nf + nlmax;
    int* rcj_s = rcp_s + nlmax;
    double* rc_s = D;                                     // D and z were consumed when ds / zs were built
    double* rs_s = z;
    __shared__ int s_nd, s_nrot;
    if (tid == 0) {
        int k = 0, nd = 0, nrot = 0, pj = -1;
        double zpj = 0.0, dpj = 0.0;
        auto step = [&](int jj, double zj, double dj) {
            if (rho * fabs(zj) <= tol) { deflpos[nd++] = jj; return; }
            if (pj < 0) { pj = jj; zpj = zj; dpj = dj; return; }
            const double tau2 = zj * zj + zpj * zpj, t = dj - dpj;
            if (fabs(t * zj * zpj) <= tol * tau2) {
                const double tau = sqrt(tau2), c_ = zj / tau, s_ = -zpj / tau;
                rcp_s[nrot] = ord[pj]; rcj_s[nrot] = ord[jj]; rc_s[nrot] = c_; rs_s[nrot] = s_; ++nrot;
                const double tt = dpj * c_ * c_ + dj * s_ * s_;
                const double dnew = dpj * s_ * s_ + dj * c_ * c_;
                ds[pj] = tt; zs[pj] = 0.0;
                deflpos[nd++] = pj;
                pj = jj; zpj = tau; dpj = dnew;
                ds[jj] = dnew; zs[jj] = tau;
            } else { polepos[k++] = pj; pj = jj; zpj = zj; dpj = dj; }
        };
        // four entries requested together (the scan only ever writes positions it has passed): one LDS round trip per four entries (eight: no further gain)
        // instead of one per entry -- the scan is a chain of such round trips (~45 ns per entry before)
        int jj = 0;
        for (; jj + 4 <= nl; jj += 4) {
            const double z0 = zs[jj], z1 = zs[jj + 1], z2 = zs[jj + 2], z3 = zs[jj + 3];
            const double d0 = ds[jj], d1 = ds[jj + 1], d2 = ds[jj + 2], d3 = ds[jj + 3];
            step(jj, z0, d0); step(jj + 1, z1, d1); step(jj + 2, z2, d2); step(jj + 3, z3, d3);
        }
        for (; jj < nl; ++jj) step(jj, zs[jj], ds[jj]);
        polepos[k++] = pj;
        s_k = k; s_nd = nd; s_nrot = nrot;
    }
    __syncthreads();
    const int k = s_k, nd = s_nd, nrot = s_nrot;
    for (int p = tid; p < k; p += 1024) { const int sp = polepos[p], c = ord[sp]; m.dl[lo + p] = ds[sp]; m.zl[lo + p] = zs[sp]; m.pcol[lo + p] = c; m.rowpole[lo + c] = p; }
    for (int t = tid; t < nd; t += 1024) { const int sp = deflpos[t]; m.dval[lo + t] = ds[sp]; m.dcol[lo + t] = ord[sp]; }
    for (int t = tid; t < nrot; t += 1024) { m.rcp[lo + t] = rcp_s[t]; m.rcj[lo + t] = rcj_s[t]; m.rc[lo + t] = rc_s[t]; m.rs[lo + t] = rs_s[t]; }
    if (tid == 0) { m.mk[lo] = k; m.mrot[lo] = nrot; m.mrho[lo] = rho; }
}

// Root j of  1 + rho sum_i z_i^2 / (d_i - lambda)  between d_j and d_{j+1} (the last one: right of d_{k-1}), one wave per root.
// The unknown is tau = lambda - d_origin with the origin at the nearer pole; the iteration interpolates the poles left of the
// root by s + a / (p - tau) and the ones right of it by r + b / (q - tau) with value and slope matched ("middle way"), and falls
// back to bisection of the bracket whenever the model's root leaves it.
__global__ void __launch_bounds__(256) dc_secular_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges, int nlmax)
{
    extern __shared__ double sh[];
    double* dl = sh;
    double* z2 = sh + nlmax;
    const DcMerge mg = merges[blockIdx.y];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, k = m.mk[lo], tid = threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x * 4 >= k) return;
    for (int i = tid; i < k; i += 256) { dl[i] = m.dl[lo + i]; const double zv = m.zl[lo + i]; z2[i] = zv * zv; }
    __syncthreads();
    const int j = blockIdx.x * 4 + (tid >> 6);
    if (j >= k) return;
    const double rho = m.mrho[lo];
    int o = 0;
    double tau;
    if (k == 1) tau = rho * z2[0];
    else {
        const bool last = j == k - 1;
        int p0, p1;
        double lo_b, hi_b;
        if (last) {
            double s = 0.0;
            for (int i = lane; i < k; i += 64) s += z2[i];
            const double width = rho * wave_sum(s), mid = 0.5 * width;
            o = j; p0 = j - 1; p1 = j;
            const double dorg = dl[o];
            double g = 0.0;
            for (int i = lane; i < k; i += 64) g += z2[i] / ((dl[i] - dorg) - mid);
            g = 1.0 + rho * wave_sum(g);
            if (g <= 0.0) { lo_b = mid; hi_b = width; } else { lo_b = 0.0; hi_b = mid; }
        } else {
            const double width = dl[j + 1] - dl[j], mid = 0.5 * width;
            p0 = j; p1 = j + 1;
            const double dj = dl[j];
            double g = 0.0;
            for (int i = lane; i < k; i += 64) g += z2[i] / ((dl[i] - dj) - mid);
            g = 1.0 + rho * wave_sum(g);
            if (g >= 0.0) { o = j; lo_b = 0.0; hi_b = mid; } else { o = j + 1; lo_b = -mid; hi_b = 0.0; }
        }
        const double dorg = dl[o];
        const double p = dl[p0] - dorg, q = dl[p1] - dorg;
        const int nlft = p0 + 1;
        tau = 0.5 * (lo_b + hi_b);
        for (int it = 0; it < 100; ++it) {
            double sg = 0.0, sa = 0.0, psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
            for (int i = lane; i < k; i += 64) {
                const double den = (dl[i] - dorg) - tau;
                const double t = rho * z2[i] / den, dt = t / den;
                sg += t; sa += fabs(t);
                if (i < nlft) { psi += t; dpsi += dt; } else { phi += t; dphi += dt; }
            }
            sg = wave_sum(sg); sa = wave_sum(sa); psi = wave_sum(psi); dpsi = wave_sum(dpsi); phi = wave_sum(phi); dphi = wave_sum(dphi);
            const double g = 1.0 + sg;
            if (fabs(g) <= DC_EPS * (8.0 * sa + 1.0)) break;
            if (g > 0.0) hi_b = tau; else lo_b = tau;
            if (hi_b - lo_b <= 2.0 * DC_EPS * fmax(fabs(lo_b), fabs(hi_b))) break;
            const double a = dpsi * (p - tau) * (p - tau), sc = psi - dpsi * (p - tau);
            const double b = dphi * (q - tau) * (q - tau), rcst = phi - dphi * (q - tau);
            const double c = 1.0 + sc + rcst;
            const double A2 = c, B2 = -(c * (p + q) + a + b), C2 = c * p * q + a * q + b * p;      // c (p-t)(q-t) + a (q-t) + b (p-t) = 0
            double t1 = 0.5 * (lo_b + hi_b), t2 = t1;
            bool h1 = false, h2 = false;
            if (A2 == 0.0) { if (B2 != 0.0) { t1 = -C2 / B2; h1 = true; } }
            else {
                const double disc = B2 * B2 - 4.0 * A2 * C2;
                if (disc >= 0.0) {
                    const double sq = sqrt(disc), qq = -0.5 * (B2 + (B2 >= 0.0 ? sq : -sq));
                    if (qq != 0.0) { t1 = C2 / qq; h1 = true; }
                    t2 = qq / A2; h2 = true;
                }
            }
            if (h1 && t1 > lo_b && t1 < hi_b) tau = t1;
            else if (h2 && t2 > lo_b && t2 < hi_b) tau = t2;
            else tau = 0.5 * (lo_b + hi_b);
        }
    }
    if (lane == 0) { m.org[lo + j] = o; m.tau[lo + j] = tau; m.lam[lo + j] = dl[o] + tau; }
}

// Loewner weights (Gu / Eisenstat: the z for which the computed roots are the exact eigenvalues -- this is what makes the
// eigenvector columns orthogonal to round-off).  k^2 work per merge: DC_SPLIT workgroups per merge, 16 lanes per pole.
constexpr int DC_SPLIT = 8;
__device__ __forceinline__ double prod16(double v)          // product over the 16 lanes of a row, in every lane
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v *= __shfl_xor(v, o, 16);
    return v;
}
__device__ __forceinline__ double sum16(double v)
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}
// |zhat_i|^2 of pole i, formed by the 16 lanes q = 0 .. 15 of a group (every lane of the group must call it)
__device__ __forceinline__ double loewner_weight(int i, int k, int q, const double* dl, const double* od, const double* tau)
{
    const double di = dl[i];
    double w = q == 0 ? (di - od[i]) - tau[i] : 1.0;
    for (int j = q; j < k; j += 16) if (j != i) w *= ((di - od[j]) - tau[j]) / (di - dl[j]);
    return prod16(w);
}
__global__ void __launch_bounds__(1024) dc_zhat_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges, int nlmax)
{
    extern __shared__ double sh[];
    double* dl = sh;
    double* od = sh + nlmax;                  // d_origin(j)
    double* tau = sh + 2 * (size_t)nlmax;
    const DcMerge mg = merges[blockIdx.y];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, k = m.mk[lo], tid = threadIdx.x;
    if ((int)blockIdx.x * 64 >= k) return;
    for (int j = tid; j < k; j += 1024) { dl[j] = m.dl[lo + j]; tau[j] = m.tau[lo + j]; }
    __syncthreads();
    for (int j = tid; j < k; j += 1024) od[j] = dl[m.org[lo + j]];
    __syncthreads();
    const int ps = tid >> 4, q = tid & 15;
    for (int i0 = blockIdx.x * 64; i0 < k; i0 += DC_SPLIT * 64) {     // (uniform trip count: the shuffles below need all 16 lanes of a pole)
        const int i = min(i0 + ps, k - 1);
        const double w = loewner_weight(i, k, q, dl, od, tau);
        if (q == 0 && i0 + ps < k) m.zhat[lo + i] = copysign(sqrt(fabs(w)), m.zl[lo + i]);
    }
}

// column norms of the merge's eigenvector block, and the final order of the node's eigenvalues.
// WITH_ZHAT (small merges, nl <= DC_FUSE_NL: one launch instead of two): every workgroup of the merge forms ALL Loewner weights of the merge
// itself first -- k^2 / 16 divisions per thread group instead of an eighth of them, which is cheaper than the ~10 us a launch costs
// before it computes anything (profiles/r05_rdm_call_kernel_sequence_m2048.txt); same arithmetic, same results as dc_zhat_kernel + this.
constexpr int DC_FUSE_NL = 384;
template <bool WITH_ZHAT>
__global__ void __launch_bounds__(1024) dc_norm_rank_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges, int nlmax)
{
    extern __shared__ double sh[];
    double* dl = sh;
    double* zh = sh + nlmax;
    double* val = sh + 2 * (size_t)nlmax;
    double* od = sh + 3 * (size_t)nlmax;      // (WITH_ZHAT only: d_origin(j) and tau_j)
    double* tau = sh + 4 * (size_t)nlmax;
    const DcMerge mg = merges[blockIdx.y];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, nl = mg.hi - mg.lo, k = m.mk[lo], tid = threadIdx.x;
    if ((int)blockIdx.x * 64 >= nl) return;
    const int ps = tid >> 4, q = tid & 15;
    for (int j = tid; j < k; j += 1024) { dl[j] = m.dl[lo + j]; if (WITH_ZHAT) tau[j] = m.tau[lo + j]; else zh[j] = m.zhat[lo + j]; }
    for (int x = tid; x < nl; x += 1024) val[x] = x < k ? m.lam[lo + x] : m.dval[lo + x - k];
    __syncthreads();
    if (WITH_ZHAT) {
        for (int j = tid; j < k; j += 1024) od[j] = dl[m.org[lo + j]];
        __syncthreads();
        for (int i0 = 0; i0 < k; i0 += 64) {                           // (uniform trip count: the shuffles need all 16 lanes of a pole)
            const int i = min(i0 + ps, k - 1);
            const double w = loewner_weight(i, k, q, dl, od, tau);
            if (q == 0 && i0 + ps < k) {
                const double zv = copysign(sqrt(fabs(w)), m.zl[lo + i]);
                zh[i] = zv;
                if (blockIdx.x == 0) m.zhat[lo + i] = zv;              // (dc_fill_u reads it)
            }
        }
        __syncthreads();
    }
    for (int j0 = blockIdx.x * 64; j0 < k; j0 += DC_SPLIT * 64) {
        const int j = min(j0 + ps, k - 1);
        const double oj = dl[m.org[lo + j]], tj = m.tau[lo + j];
        double s = 0.0;
        for (int i = q; i < k; i += 16) { const double t = zh[i] / ((dl[i] - oj) - tj); s += t * t; }
        s = sum16(s);
        if (q == 0 && j0 + ps < k) m.cnorm[lo + j] = 1.0 / sqrt(s);
    }
    for (int x0 = blockIdx.x * 64; x0 < nl; x0 += DC_SPLIT * 64) {
        const int x = min(x0 + ps, nl - 1);
        const double v = val[x];
        int r = 0;
        for (int p = q; p < nl; p += 16) { const double u = val[p]; r += (u < v) || (u == v && p < x); }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) r += __shfl_xor(r, o, 16);
        if (q == 0 && x0 + ps < nl) {
            m.dcur[lo + r] = v;
            m.colroot[lo + r] = x < k ? x : -(x - k) - 1;
            if (x >= k) m.pcolmap[lo + m.dcol[lo + x - k]] = r;
        }
    }
}

// U (node block): column of root j = zhat_i / (d_i - lambda_j), normalised, in the rows of the non-deflated poles; column of a
// deflated pole = the unit vector of its old column.  64 x 64 elements per workgroup.
__global__ void __launch_bounds__(256) dc_fill_u_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges)
{
    __shared__ int rp[64];
    __shared__ double rz[64], rd[64];
    const DcMerge mg = merges[blockIdx.z];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, nl = mg.hi - mg.lo, tid = threadIdx.x;
    const int c = blockIdx.x * 64 + (tid & 63), rbase = blockIdx.y * 64;
    if ((int)blockIdx.x * 64 >= nl || rbase >= nl) return;
    if (tid < 64) {
        const int r = rbase + tid;
        int i = -1;
        if (r < nl) i = m.rowpole[lo + r];
        rp[tid] = i;
        if (i >= 0) { rz[tid] = m.zhat[lo + i]; rd[tid] = m.dl[lo + i]; }
    }
    __syncthreads();
    if (c >= nl) return;
    const int cr = m.colroot[lo + c];
    double od = 0.0, tj = 0.0, cn = 0.0;
    int dcolv = -1;
    if (cr >= 0) { od = m.dl[lo + m.org[lo + cr]]; tj = m.tau[lo + cr]; cn = m.cnorm[lo + cr]; }
    else dcolv = m.dcol[lo - cr - 1];
    for (int rr = tid >> 6; rr < 64; rr += 4) {
        const int r = rbase + rr;
        if (r >= nl) break;
        double v;
        if (cr >= 0) v = rp[rr] >= 0 ? rz[rr] / ((rd[rr] - od) - tj) * cn : 0.0;
        else v = dcolv == r ? 1.0 : 0.0;
        m.U[(int64_t)(lo + r) * m.ldu + lo + c] = v;
    }
}

// The type-2 deflation rotations act on columns of blockdiag(Q1, Q2); Q G U' = Q (G U'), so they are applied to the ROWS of U
// instead, in reverse order, one thread per column of U.  A chain pj_1 -> jj_1 = pj_2 -> jj_2 .. is a register recurrence: the
// row of a deflated pole is still the unit row of its final column when its rotation is reached, so nothing but the survivor's
// row is ever loaded.
__global__ void __launch_bounds__(256) dc_rot_kernel(const DcMat* __restrict__ mats, const DcMerge* __restrict__ merges, int nlmax)
{
    extern __shared__ double sh[];
    double* rc = sh;
    double* rs = sh + nlmax;
    int* rcp = (int*)(sh + 2 * (size_t)nlmax);
    int* rcj = rcp + nlmax;
    int* pmap = rcj + nlmax;
    const DcMerge mg = merges[blockIdx.y];
    const DcMat m = mats[mg.mat];
    const int lo = mg.lo, nl = mg.hi - mg.lo, nrot = m.mrot[lo], tid = threadIdx.x;
    if (nrot == 0 || (int)blockIdx.x * 256 >= nl) return;
    for (int t = tid; t < nrot; t += 256) { rc[t] = m.rc[lo + t]; rs[t] = m.rs[lo + t]; const int cp = m.rcp[lo + t]; rcp[t] = cp; rcj[t] = m.rcj[lo + t]; pmap[t] = m.pcolmap[lo + cp]; }
    __syncthreads();
    const int p = blockIdx.x * 256 + tid;
    if (p >= nl) return;
    double* Ucol = m.U + (int64_t)lo * m.ldu + lo + p;
    int t = nrot - 1;
    while (t >= 0) {
        double R = Ucol[(int64_t)rcj[t] * m.ldu];
        bool cont;
        do {
            const double e = pmap[t] == p ? 1.0 : 0.0, c = rc[t], s = rs[t];
            Ucol[(int64_t)rcj[t] * m.ldu] = s * e + c * R;
            R = c * e - s * R;
            cont = t > 0 && rcj[t - 1] == rcp[t];
            if (!cont) Ucol[(int64_t)rcp[t] * m.ldu] = R;
            --t;
        } while (cont);
    }
}

__global__ void __launch_bounds__(256) dc_out_kernel(const DcMat* __restrict__ mats)
{
    const DcMat m = mats[blockIdx.y];
    const double sc = m.scale[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m.n; i += gridDim.x * 256) m.w[i] = m.dcur[i] * sc;
}

// boundaries of the tree's nodes at a depth: repeated halving of [0, n)
std::vector<int> tree_bounds(int n, int level)
{
    std::vector<int> b{0, n};
    for (int l = 0; l < level; ++l) {
        std::vector<int> nb;
        for (size_t i = 0; i + 1 < b.size(); ++i) { nb.push_back(b[i]); nb.push_back((b[i] + b[i + 1]) / 2); }
        nb.push_back(n);
        b.swap(nb);
    }
    return b;
}

bool g_coop_disabled = false;                       // set once a persistent round has timed out (process-wide)
int32_t g_coop_timeouts = 0;                        // how often that happened

struct GemmSet {                                   // the tile lists of one dependent GEMM step
    size_t big_off = 0, small_off = 0;
    int32_t nbig = 0, nsmall = 0;
};

template <class K> dmrgx_status set_dyn_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) DMRGX_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DMRGX_OK;
}

}  // namespace

void symeig_set_persistent(bool on) { g_coop_disabled = !on; }
void symeig_note_timeout(int32_t status)
{
    if (!g_coop_disabled) fprintf(stderr, status == 2 ? "[dmrgx] persistent tridiagonalisation: a workgroup was lapped by its partners (protocol error): using one launch per column from now on\n"
                                                       : "[dmrgx] persistent tridiagonalisation timed out waiting for a partner workgroup (GPU shared with another "
                                                         "persistent kernel?): using one launch per column from now on\n");
    g_coop_disabled = true;
    ++g_coop_timeouts;
}
int32_t symeig_deferred_timed_out(const SymEigDeferred& d) { return d.status_host ? *d.status_host : 0; }
void symeig_process_state(int32_t* timeouts, int32_t* persistent_off)
{
    if (timeouts) *timeouts = g_coop_timeouts;
    if (persistent_off) *persistent_off = g_coop_disabled ? 1 : 0;
}

dmrgx_status symeig_batched(const std::vector<SymEigMat>& mats_in, hipStream_t st, SymEigReport* report, SymEigDeferred* deferred)
{
    SymEigReport rep_local;
    SymEigReport& rep = report ? *report : rep_local;
    rep = SymEigReport();
    if (deferred) deferred->status_host = nullptr;
    std::vector<SymEigMat> M;
    for (const SymEigMat& s : mats_in) {
        if (s.n < 0 || s.n > SYMEIG_MAX_N) DMRGX_FAIL(DMRGX_ERR_ARG, "symeig: matrix of order %d (supported: 0..%d)", s.n, SYMEIG_MAX_N);
        if (s.n > 0 && (!s.A || !s.X || !s.w || s.lda < s.n || s.ldx < s.n)) DMRGX_FAIL(DMRGX_ERR_ARG, "symeig: bad matrix descriptor");
        if (s.n > 0) M.push_back(s);
    }
    const int nm = (int)M.size();
    if (nm == 0) return DMRGX_OK;
    // developer aid (DMRGX_SYMEIG_TRACE): host clock at the milestones of the call, no synchronisation -- where the HOST spends its time
    static const bool host_trace = getenv("DMRGX_SYMEIG_TRACE") != nullptr;
    const auto ht0 = std::chrono::steady_clock::now();
    std::vector<std::pair<const char*, double>> hmarks;
    auto hmark = [&](const char* what) { if (host_trace) hmarks.push_back({what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count()}); };
    int nmax = 0;
    for (const SymEigMat& s : M) nmax = std::max(nmax, s.n);

    // ---- workspace -----------------------------------------------------------------------------------------------------
    struct Ws { int64_t VT, Vc, TV, Q1, U, y, d, e, tau, dcur, scale, vecs, G, Tn, W; int64_t ints; int nblk; };
    std::vector<Ws> ws(nm);
    int64_t dtot = 0, itot = 0;
    constexpr int NVEC = 10, NINT = 10;              // per-position double / int arrays of DcMat
    for (int i = 0; i < nm; ++i) {
        const int64_t n = M[i].n, nn = n * n;
        Ws& w = ws[i];
        w.nblk = (int)((std::max<int64_t>(n - 2, 0) + WY_NB - 1) / WY_NB);
        w.VT = dtot; dtot += nn;
        w.Vc = dtot; dtot += nn;
        w.TV = dtot; dtot += nn;
        w.Q1 = dtot; dtot += nn;
        w.U = dtot; dtot += nn;
        w.y = dtot; dtot += 2 * n;
        w.d = dtot; dtot += n; w.e = dtot; dtot += n; w.tau = dtot; dtot += n; w.dcur = dtot; dtot += n;
        w.scale = dtot; dtot += 8;
        w.vecs = dtot; dtot += NVEC * n;
        w.G = dtot; dtot += (int64_t)w.nblk * WY_NB * WY_NB;
        w.Tn = dtot; dtot += (int64_t)w.nblk * WY_NB * WY_NB;
        w.W = dtot; dtot += (int64_t)WY_NB * n;
        w.ints = itot; itot += NINT * n;
    }
    DevBuf dbuf, ibuf;
    DMRGX_CHK(dbuf.alloc((size_t)dtot * sizeof(double)));
    DMRGX_CHK(ibuf.alloc((size_t)itot * sizeof(int32_t)));
    double* B = dbuf.as<double>();
    int32_t* I = ibuf.as<int32_t>();
    hmark("workspace");

    // ---- 1. tridiagonalisation -------------------------------------------------------------------------------------------
    // Matrices whose rows fit the LDS of at most all CUs go through the persistent kernel (in rounds of <= #CUs workgroups and
    // <= 32 matrices); larger ones, and every one of them if a round reports a bounded-spin timeout, through one launch per column.
    auto trid_by_launches = [&](const std::vector<int>& set) -> dmrgx_status {
        for (size_t c0 = 0; c0 < set.size(); c0 += TRID_MAXM) {
            const int cn = (int)std::min<size_t>(TRID_MAXM, set.size() - c0);
            TridArgs ta;
            int gmax = 0;
            for (int i = 0; i < TRID_MAXM; ++i) {
                if (i < cn) { const int q = set[c0 + i]; ta.m[i] = TridMat{M[q].A, B + ws[q].VT, B + ws[q].y, B + ws[q].d, B + ws[q].e, B + ws[q].tau, M[q].n, M[q].lda, M[q].n, 0}; gmax = std::max(gmax, M[q].n); }
                else ta.m[i] = TridMat{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
            }
            const size_t lds = (size_t)3 * gmax * sizeof(double);
            DMRGX_CHK(set_dyn_lds(trid_step_kernel, lds));
            const unsigned gx = (unsigned)((gmax + TRID_ROWS - 1) / TRID_ROWS);
            for (int j = 0; j < gmax; ++j) hipLaunchKernelGGL(trid_step_kernel, dim3(gx, (unsigned)cn), dim3(TRID_THREADS), lds, st, ta, j);
        }
        DMRGX_HIP(hipGetLastError());
        return DMRGX_OK;
    };
    static const bool coop_wanted = !(getenv("DMRGX_TRID") && std::string(getenv("DMRGX_TRID")) == "launch");     // developer aid / A-B
    std::vector<int> launch_set, coop_set;
    DevBuf gran;                                       // status word (first 16 bytes) + granule buffers of the persistent rounds
    int32_t coop_status = 0;
    bool coop_ran = false;
    if (coop_wanted && !g_coop_disabled) {
        int dev = 0, ncu = 0;
        DMRGX_HIP(hipGetDevice(&dev));
        DMRGX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        int max_lds = 0;
        DMRGX_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        const int64_t dyn_max = std::min<int64_t>(max_lds, 160 * 1024) - 256;       // (static LDS of the kernel: a few words)
        std::vector<int> order(nm);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return M[a].n > M[b].n; });
        struct Round { std::vector<int> mats, G; int wgs = 0; };
        std::vector<Round> rounds;
        // Workgroups per matrix: at least what LDS capacity asks for.  CUs that this leaves idle go, one at a time, to the matrix with the
        // longest estimated chain -- n columns x (a fixed part + the workgroup's own row pass, which is proportional to rows per workgroup
        // x row length) -- down to `rows_floor` rows per workgroup: the launch lasts as long as its longest chain, a workgroup's row pass
        // is one link of every column of it, and among few workgroups more partners cost the all-gather almost nothing.  (Small m: a dozen
        // matrices of n <= 300 on 256 CUs go from 75 to 8-16 rows per workgroup: 888 -> 672 us per call at m = 512.)
        constexpr int rows_floor = 8;      // (4 and 2 rows per workgroup measured flat, round 3)
        std::vector<int> Gmin(nm, 0), Gof(nm, 0);
        {
            int64_t used = 0;
            for (int q = 0; q < nm; ++q) {
                const int64_t n = M[q].n, nv = (n + 1) & ~(int64_t)1, cap = (dyn_max / 8 - TC_VECS * nv) / nv;      // rows of this matrix one workgroup can hold
                Gmin[q] = cap >= 1 ? (int)((n + cap - 1) / cap) : ncu + 1;
                Gof[q] = Gmin[q];
                if (Gmin[q] <= ncu) used += Gmin[q];
            }
            // estimated chain [us]: per column 2.7 fixed (exchange, vector work, barriers) + 1.35 for 18 rows of 1037 columns (measured, m = 2048)
            auto chain = [&](int q) { const double n = M[q].n, rows = std::ceil(n / Gof[q]); return n * (2.7 + 1.35 * (rows / 18.0) * (n / 1037.0)); };
            // Only when the largest matrix is itself small: at m = 2048 (n up to 1037, G = 58 by capacity) more workgroups made the call
            // SLOWER on the same box -- 8.49 ms per truncation with none, 8.60 with the spare CUs on the small matrices, 8.62 with them on
            // the largest -- more partners lengthen the exchange there by more than the shorter row pass saves.
            if (used <= ncu && nmax <= 400) {          // (m = 1024, n up to ~520: no difference either way)
                for (int64_t spare = ncu - used; spare > 0; --spare) {
                    int best = -1; double tbest = 0;
                    for (int q = 0; q < nm; ++q) {
                        if (Gmin[q] > ncu || (M[q].n + Gof[q] - 1) / Gof[q] <= rows_floor) continue;
                        const double t = chain(q);
                        if (best < 0 || t > tbest) { best = q; tbest = t; }
                    }
                    if (best < 0) break;
                    ++Gof[best];
                }
            }
        }
        for (int q : order) {
            const int G = Gof[q];
            if (G > ncu) { launch_set.push_back(q); continue; }
            Round* r = nullptr;
            for (Round& c : rounds) if (c.wgs + G <= ncu && (int)c.mats.size() < TRID_MAXM) { r = &c; break; }
            if (!r) { rounds.emplace_back(); r = &rounds.back(); }
            r->mats.push_back(q); r->G.push_back(G); r->wgs += G;
            coop_set.push_back(q);
            rep.max_workgroups_per_matrix = std::max(rep.max_workgroups_per_matrix, G);
        }
        if (!rounds.empty()) {
            int64_t gtot = 2;                                                        // in u64 words; the status word owns the first 16 bytes
            std::vector<int64_t> goff(nm, 0);
            for (int q : coop_set) { goff[q] = gtot; gtot += 8 * (int64_t)M[q].n; }   // ybuf, rowbuf: 2 parities x n x 2 granules each
            gtot = (gtot + 1) & ~(int64_t)1;
            DMRGX_CHK(gran.alloc((size_t)gtot * sizeof(u64)));
            DMRGX_HIP(hipMemsetAsync(gran.p, 0, (size_t)gtot * sizeof(u64), st));     // epochs start at 1: a zeroed granule is "not yet"
            u64* GB = gran.as<u64>();
            hmark("granules");
            DMRGX_HIP(hipFuncSetAttribute((const void*)trid_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
            hmark("func attribute");
            for (const Round& r : rounds) {
                TcArgs ta;
                ta.status = reinterpret_cast<int32_t*>(GB); ta.nm = (int32_t)r.mats.size();
                // test hook (tests/test_gpu_kron.py: the time-out -> launch-per-column transition, once): the first persistent round of the process
                // loses one publication
                static bool fault_pending = getenv("DMRGX_TRID_FAULT") != nullptr;
                ta.fault = fault_pending ? 1 : 0; fault_pending = false;
                ta.prof = getenv("DMRGX_TRID_PROF") ? reinterpret_cast<long long*>(B + ws[r.mats[0]].y) : nullptr;      // (the y scratch of the launch path is idle here)
                size_t lds = 0;
                int wg0 = 0;
                for (int i = 0; i < TRID_MAXM; ++i) {
                    if (i < (int)r.mats.size()) {
                        const int q = r.mats[(size_t)i], n = M[q].n, G = r.G[(size_t)i];
                        ta.m[i] = TcMat{M[q].A, B + ws[q].VT, B + ws[q].d, B + ws[q].e, B + ws[q].tau, GB + goff[q], GB + goff[q] + 4 * (int64_t)n, n, M[q].lda, n, G, wg0, 0};
                        wg0 += G;
                        lds = std::max(lds, (size_t)(TC_VECS + (n + G - 1) / G) * (size_t)((n + 1) & ~1) * sizeof(double));
                    } else ta.m[i] = TcMat{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 1 << 30, 0};
                }
                hipLaunchKernelGGL(trid_coop_kernel, dim3((unsigned)wg0), dim3(TC_THREADS), lds, st, ta);
                DMRGX_HIP(hipGetLastError());
            }
            coop_ran = true;
        }
    } else for (int q = 0; q < nm; ++q) launch_set.push_back(q);
    if (!launch_set.empty()) DMRGX_CHK(trid_by_launches(launch_set));
    rep.persistent_matrices = (int32_t)coop_set.size(); rep.launch_matrices = (int32_t)launch_set.size();
    hmark("trid launched");

    // ---- 3a. (independent of the eigenvectors) V = (V^T)^T, the Gram blocks, T factors and T V^T ---------------------------------
    std::vector<GProd> prods;
    std::vector<GGroup> groups;
    std::vector<GTile> tiles;                         // all tile lists, one after the other
    auto add_set = [&](std::vector<GTile>& big, std::vector<GTile>& small) {
        GemmSet s;
        ggemm_schedule(big, groups, 2); ggemm_schedule(small, groups);
        s.big_off = tiles.size(); s.nbig = (int32_t)big.size(); tiles.insert(tiles.end(), big.begin(), big.end());
        s.small_off = tiles.size(); s.nsmall = (int32_t)small.size(); tiles.insert(tiles.end(), small.begin(), small.end());
        return s;
    };
    auto add_gemm = [&](std::vector<GTile>& big, std::vector<GTile>& small, double* C, int ldc, int Mr, int Nc, const double* A, int lda, const double* Bm, int ldb, int K, int accumulate) {
        if (Mr <= 0 || Nc <= 0 || K <= 0) return;
        prods.push_back(GProd{A, Bm, lda, ldb, K, GPROD_GEMM, 1.0});
        groups.push_back(GGroup{C, ldc, Mr, Nc, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, accumulate});
        ggemm_append_tiles_mixed(big, small, (int32_t)groups.size() - 1, Mr, Nc, (K + GG_BK - 1) / GG_BK);
    };
    std::vector<SqPair> tp(nm);
    std::vector<WyBlock> wyb;
    GemmSet set_gram, set_tv, set_t12a, set_t12b;
    int max_nblk = 0;
    {
        std::vector<GTile> gb, gs, tb, tsm, pb_, ps_, qb_, qs_;
        for (int i = 0; i < nm; ++i) {
            const int n = M[i].n;
            const Ws& w = ws[i];
            max_nblk = std::max(max_nblk, w.nblk);
            tp[i] = SqPair{B + w.VT, B + w.Vc, n, n, n, 0};
            for (int b = 0; b < w.nblk; ++b) {
                const int b0 = b * WY_NB, kb = std::min(WY_NB, n - 2 - b0), r0 = b0 + 1;      // reflectors b0 .. b0+kb-1 live in rows >= b0+1
                double* G = B + w.G + (int64_t)b * WY_NB * WY_NB;
                double* Tn = B + w.Tn + (int64_t)b * WY_NB * WY_NB;
                add_gemm(gb, gs, G, WY_NB, kb, kb, B + w.VT + (int64_t)b0 * n + r0, n, B + w.Vc + (int64_t)r0 * n + b0, n, n - r0, 0);      // G = V_b^T V_b
                const int k1 = std::min(kb, WY_SUB), k2 = kb - k1;
                wyb.push_back(WyBlock{G, B + w.tau + b0, Tn, k1, WY_NB, k2, 0});
                if (k2 > 0) {
                    wyb.push_back(WyBlock{G + (int64_t)WY_SUB * WY_NB + WY_SUB, B + w.tau + b0 + WY_SUB, Tn + (int64_t)WY_SUB * WY_NB + WY_SUB, k2, WY_NB, 0, 0});
                    // Tneg12 = Tneg1 . G12 . Tneg2 (= -T12): P = G12 . Tneg2 parked in the unused lower-left block of G, then Tneg12 = Tneg1 . P
                    double* Pm = G + (int64_t)WY_SUB * WY_NB;
                    add_gemm(pb_, ps_, Pm, WY_NB, k1, k2, G + WY_SUB, WY_NB, Tn + (int64_t)WY_SUB * WY_NB + WY_SUB, WY_NB, k2, 0);
                    add_gemm(qb_, qs_, Tn + WY_SUB, WY_NB, k1, k2, Tn, WY_NB, Pm, WY_NB, k1, 0);
                }
                add_gemm(tb, tsm, B + w.TV + (int64_t)b0 * n, n, kb, n, Tn, WY_NB, B + w.VT + (int64_t)b0 * n, n, kb, 0);     // (T V^T)_b = Tneg_b . V_b^T
            }
        }
        set_t12a = add_set(pb_, ps_);
        set_t12b = add_set(qb_, qs_);
        set_gram = add_set(gb, gs);
        set_tv = add_set(tb, tsm);
    }

    // ---- 2. divide-and-conquer tables --------------------------------------------------------------------------------------
    std::vector<DcMat> dm(nm);
    std::vector<DcLeaf> leaves;
    std::vector<int> depth(nm, 0);
    int dmax = 0;
    for (int i = 0; i < nm; ++i) {
        const int n = M[i].n;
        const Ws& w = ws[i];
        int D = 0;
        while ((n + (1 << D) - 1) / (1 << D) > DC_LEAF) ++D;
        depth[i] = D; dmax = std::max(dmax, D);
        rep.merge_levels = dmax; rep.wy_blocks_max = std::max(rep.wy_blocks_max, w.nblk);
        DcMat& q = dm[i];
        q.Q[0] = M[i].X; q.ldq[0] = M[i].ldx;
        q.Q[1] = B + w.Q1; q.ldq[1] = n;
        q.U = B + w.U; q.ldu = n;
        q.d = B + w.d; q.e = B + w.e; q.dcur = B + w.dcur; q.scale = B + w.scale; q.w = M[i].w;
        double* v = B + w.vecs;
        q.dl = v; q.zl = v + n; q.tau = v + 2 * (int64_t)n; q.lam = v + 3 * (int64_t)n; q.zhat = v + 4 * (int64_t)n; q.cnorm = v + 5 * (int64_t)n;
        q.dval = v + 6 * (int64_t)n; q.rc = v + 7 * (int64_t)n; q.rs = v + 8 * (int64_t)n; q.mrho = v + 9 * (int64_t)n;
        int32_t* iv = I + w.ints;
        q.org = iv; q.pcol = iv + n; q.dcol = iv + 2 * (int64_t)n; q.rowpole = iv + 3 * (int64_t)n; q.colroot = iv + 4 * (int64_t)n; q.pcolmap = iv + 5 * (int64_t)n;
        q.rcp = iv + 6 * (int64_t)n; q.rcj = iv + 7 * (int64_t)n; q.mk = iv + 8 * (int64_t)n; q.mrot = iv + 9 * (int64_t)n;
        q.n = n;
        const std::vector<int> b = tree_bounds(n, D);
        for (size_t t = 0; t + 1 < b.size(); ++t) leaves.push_back(DcLeaf{i, b[t], b[t + 1], D & 1});
    }
    struct Step { size_t merge_off = 0; int nmerge = 0, nlmax = 0; GemmSet gemm; };
    std::vector<Step> steps((size_t)dmax);
    std::vector<DcMerge> merges;
    for (int t = 1; t <= dmax; ++t) {
        Step& s = steps[(size_t)t - 1];
        s.merge_off = merges.size();
        std::vector<GTile> gb, gs;
        for (int i = 0; i < nm; ++i) {
            if (depth[i] < t) continue;
            const int lev = depth[i] - t, n = M[i].n;           // the level being produced; its children live at lev + 1
            const std::vector<int> pb = tree_bounds(n, lev), cb = tree_bounds(n, lev + 1);
            const int src = (lev + 1) & 1, dst = lev & 1;
            for (size_t u = 0; u + 1 < pb.size(); ++u) {
                const int lo = pb[u], mid = cb[2 * u + 1], hi = pb[u + 1];
                merges.push_back(DcMerge{i, lo, mid, hi, src, 0});
                s.nlmax = std::max(s.nlmax, hi - lo);
                if (deferred && lev == 0) continue;                 // the root's GEMM: symeig_finish, for the kept columns only
                double* Qd = dm[i].Q[dst]; const int ldd = dm[i].ldq[dst];
                const double* Qs = dm[i].Q[src]; const int lds_ = dm[i].ldq[src];
                const double* U = dm[i].U; const int ldu = dm[i].ldu;
                add_gemm(gb, gs, Qd + (int64_t)lo * ldd + lo, ldd, mid - lo, hi - lo, Qs + (int64_t)lo * lds_ + lo, lds_, U + (int64_t)lo * ldu + lo, ldu, mid - lo, 0);
                add_gemm(gb, gs, Qd + (int64_t)mid * ldd + lo, ldd, hi - mid, hi - lo, Qs + (int64_t)mid * lds_ + mid, lds_, U + (int64_t)mid * ldu + lo, ldu, hi - mid, 0);
            }
        }
        s.nmerge = (int)(merges.size() - s.merge_off);
        s.gemm = add_set(gb, gs);
    }

    // ---- 3b. back-transformation steps (last block first), aligned at the end of every matrix's block list ------------------------
    std::vector<GemmSet> bt_w((size_t)max_nblk), bt_x((size_t)max_nblk);
    for (int s = 0; s < max_nblk && !deferred; ++s) {
        std::vector<GTile> wb, wsm, xb, xs;
        for (int i = 0; i < nm; ++i) {
            const Ws& w = ws[i];
            const int b = w.nblk - 1 - s, n = M[i].n;
            if (b < 0) continue;
            const int b0 = b * WY_NB, kb = std::min(WY_NB, n - 2 - b0), r0 = b0 + 1;
            double* X = M[i].X; const int ldx = M[i].ldx;
            add_gemm(wb, wsm, B + w.W, n, kb, n, B + w.TV + (int64_t)b0 * n + r0, n, X + (int64_t)r0 * ldx, ldx, n - r0, 0);          // W = -(T V^T) X
            add_gemm(xb, xs, X + (int64_t)r0 * ldx, ldx, n - r0, n, B + w.Vc + (int64_t)r0 * n + b0, n, B + w.W, n, kb, 1);             // X += V W
        }
        bt_w[(size_t)s] = add_set(wb, wsm);
        bt_x[(size_t)s] = add_set(xb, xs);
    }

    hmark("tables built");
    // ---- uploads ---------------------------------------------------------------------------------------------------------------
    if (prods.empty()) prods.push_back(GProd{nullptr, nullptr, 0, 0, 0, GPROD_GEMM, 0.0});
    if (groups.empty()) groups.push_back(GGroup{nullptr, 0, 0, 0, 0, 0, 0, 0});
    if (tiles.empty()) tiles.push_back(GTile{-1, 0, 0, 0});
    if (merges.empty()) merges.push_back(DcMerge{0, 0, 0, 0, 0, 0});
    if (wyb.empty()) wyb.push_back(WyBlock{nullptr, nullptr, nullptr, 0, 0, 0, 0});
    DevBuf d_tab;                                      // all eight tables in one copy
    PackedUpload pk;
    const size_t o_prods = pk.add(prods), o_groups = pk.add(groups), o_tiles = pk.add(tiles), o_tp = pk.add(tp), o_wyb = pk.add(wyb), o_dm = pk.add(dm),
                 o_leaves = pk.add(leaves), o_merges = pk.add(merges);
    DMRGX_CHK(pk.upload(d_tab, st));
    const GProd* d_prods = packed_at<GProd>(d_tab, o_prods);
    const GGroup* d_groups = packed_at<GGroup>(d_tab, o_groups);
    const GTile* d_tiles = packed_at<GTile>(d_tab, o_tiles);
    const SqPair* d_tp = packed_at<SqPair>(d_tab, o_tp);
    const WyBlock* d_wyb = packed_at<WyBlock>(d_tab, o_wyb);
    const DcMat* d_dm = packed_at<DcMat>(d_tab, o_dm);
    const DcLeaf* d_leaves = packed_at<DcLeaf>(d_tab, o_leaves);
    const DcMerge* d_merges = packed_at<DcMerge>(d_tab, o_merges);
    hmark("tables uploaded");
    // ---- the persistent rounds have had the host's table building and the table uploads to run in: did every workgroup get its partners? ---------------
    if (coop_ran && getenv("DMRGX_TRID_PROF")) {
        long long pr[4];
        DMRGX_HIP(hipMemcpy(pr, B + ws[coop_set[0]].y, sizeof(pr), hipMemcpyDeviceToHost));
        fprintf(stderr, "[trid-prof] n=%d: wave 0 of workgroup 1, mean per column [us]: consume %.2f  vector work %.2f  own rows %.2f  end barrier %.2f\n", M[coop_set[0]].n, pr[0] / 100.0, pr[1] / 100.0, pr[2] / 100.0, pr[3] / 100.0);
    }
    if (coop_ran && deferred) {
        // Two-phase call: the status word travels to pinned memory behind the persistent rounds and is looked at by the CALLER behind its own
        // synchronisation (the spectra's) -- symeig_deferred_timed_out -- so that everything below is queued while the tridiagonalisation still
        // runs: the divide and conquer starts the moment it ends instead of one host hand-over later.  A time-out (never seen outside the
        // test hook and GPUs shared with another persistent kernel) then costs the caller a second call, on the launch path.
        static thread_local int32_t* pin_status = nullptr;
        if (!pin_status) DMRGX_HIP(hipHostMalloc((void**)&pin_status, 64, hipHostMallocDefault));
        *pin_status = 0;
        DMRGX_HIP(hipMemcpyAsync(pin_status, gran.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        deferred->status_host = pin_status;
    } else if (coop_ran) {
        DMRGX_HIP(hipMemcpyAsync(&coop_status, gran.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));      // (a pageable copy: it may block, so it is issued here)
        DMRGX_HIP(hipStreamSynchronize(st));
        hmark("trid done (sync)");
        if (coop_status != 0) {
            symeig_note_timeout(coop_status);
            rep.timed_out = coop_status; rep.launch_matrices += rep.persistent_matrices; rep.persistent_matrices = 0;
            DMRGX_CHK(trid_by_launches(coop_set));     // A was only read by the persistent kernel
        }
    }

    auto run_set = [&](const GemmSet& s) -> dmrgx_status {
        DMRGX_CHK(ggemm_launch(d_tiles + s.big_off, d_groups, d_prods, s.nbig, st, 1));
        DMRGX_CHK(ggemm_launch(d_tiles + s.small_off, d_groups, d_prods, s.nsmall, st, 0));
        return DMRGX_OK;
    };

    // ---- 3a: launches ----------------------------------------------------------------------------------------------------------
    // The WY factors (V^T, Gram matrix, T^-1 by one latency-bound workgroup per block, T12, T V^T) and the divide and conquer below both
    // need the tridiagonalisation only, and neither fills the chip: the WY chain goes to a second stream and joins in front of the
    // back-transformation (m = 512: ~100 us per call; timeline in profiles/r04_rdm_timeline_m512.txt).
    const bool any_blk = max_nblk > 0;
    static thread_local hipStream_t side = nullptr;            // (one per host thread: callers on different threads / devices do not share it)
    static thread_local hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (any_blk) {
        if (!side) {
            DMRGX_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            DMRGX_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            DMRGX_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        DMRGX_HIP(hipEventRecord(ev_fork, st));
        DMRGX_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        auto run_set_side = [&](const GemmSet& s) -> dmrgx_status {
            DMRGX_CHK(ggemm_launch(d_tiles + s.big_off, d_groups, d_prods, s.nbig, side, 1));
            DMRGX_CHK(ggemm_launch(d_tiles + s.small_off, d_groups, d_prods, s.nsmall, side, 0));
            return DMRGX_OK;
        };
        const unsigned t32 = (unsigned)((nmax + 31) / 32);
        hipLaunchKernelGGL(transpose_sq_kernel, dim3(t32, t32, (unsigned)nm), dim3(256), 0, side, d_tp);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(run_set_side(set_gram));
        hipLaunchKernelGGL(wy_tinv_kernel, dim3((unsigned)wyb.size()), dim3(WY_SUB), 0, side, d_wyb);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(run_set_side(set_t12a));
        DMRGX_CHK(run_set_side(set_t12b));
        DMRGX_CHK(run_set_side(set_tv));
        DMRGX_HIP(hipEventRecord(ev_join, side));
    }

    // ---- 2: launches -------------------------------------------------------------------------------------------------------------
    const DcMat* ddm = d_dm;
    hipLaunchKernelGGL(dc_scale_kernel, dim3((unsigned)nm), dim3(256), 0, st, ddm);
    hipLaunchKernelGGL(dc_leaf_kernel, dim3((unsigned)leaves.size()), dim3(DC_LEAF_THREADS), 0, st, ddm, d_leaves);
    DMRGX_HIP(hipGetLastError());
    for (const Step& s : steps) {
        if (s.nmerge == 0) continue;
        const DcMerge* mp = d_merges + s.merge_off;
        const int nl = s.nlmax;
        const size_t lds_defl = (size_t)nl * (4 * sizeof(double) + 5 * sizeof(int)) + 16;
        const size_t lds_sec = (size_t)nl * 2 * sizeof(double);
        const size_t lds_wgt = (size_t)nl * 3 * sizeof(double);
        const size_t lds_rot = (size_t)nl * (2 * sizeof(double) + 3 * sizeof(int)) + 16;
        DMRGX_CHK(set_dyn_lds(dc_deflate_kernel, lds_defl)); DMRGX_CHK(set_dyn_lds(dc_secular_kernel, lds_sec));
        const bool fuse_wgt = nl <= DC_FUSE_NL;
        const size_t lds_fused = (size_t)nl * 5 * sizeof(double);
        if (fuse_wgt) DMRGX_CHK(set_dyn_lds(dc_norm_rank_kernel<true>, lds_fused));
        else { DMRGX_CHK(set_dyn_lds(dc_zhat_kernel, lds_wgt)); DMRGX_CHK(set_dyn_lds(dc_norm_rank_kernel<false>, lds_wgt)); }
        DMRGX_CHK(set_dyn_lds(dc_rot_kernel, lds_rot));
        hipLaunchKernelGGL(dc_deflate_kernel, dim3((unsigned)s.nmerge), dim3(1024), lds_defl, st, ddm, mp, nl);
        hipLaunchKernelGGL(dc_secular_kernel, dim3((unsigned)((nl + 3) / 4), (unsigned)s.nmerge), dim3(256), lds_sec, st, ddm, mp, nl);
        if (fuse_wgt) hipLaunchKernelGGL(dc_norm_rank_kernel<true>, dim3(DC_SPLIT, (unsigned)s.nmerge), dim3(1024), lds_fused, st, ddm, mp, nl);
        else {
            hipLaunchKernelGGL(dc_zhat_kernel, dim3(DC_SPLIT, (unsigned)s.nmerge), dim3(1024), lds_wgt, st, ddm, mp, nl);
            hipLaunchKernelGGL(dc_norm_rank_kernel<false>, dim3(DC_SPLIT, (unsigned)s.nmerge), dim3(1024), lds_wgt, st, ddm, mp, nl);
        }
        hipLaunchKernelGGL(dc_fill_u_kernel, dim3((unsigned)((nl + 63) / 64), (unsigned)((nl + 63) / 64), (unsigned)s.nmerge), dim3(256), 0, st, ddm, mp);
        hipLaunchKernelGGL(dc_rot_kernel, dim3((unsigned)((nl + 255) / 256), (unsigned)s.nmerge), dim3(256), lds_rot, st, ddm, mp, nl);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(run_set(s.gemm));
    }
    hipLaunchKernelGGL(dc_out_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)nm), dim3(256), 0, st, ddm);
    DMRGX_HIP(hipGetLastError());

    // ---- 3b: launches ------------------------------------------------------------------------------------------------------------
    if (any_blk) DMRGX_HIP(hipStreamWaitEvent(st, ev_join, 0));
    for (int s = 0; s < max_nblk && !deferred; ++s) { DMRGX_CHK(run_set(bt_w[(size_t)s])); DMRGX_CHK(run_set(bt_x[(size_t)s])); }
    if (deferred) {
        // everything symeig_finish needs: the matrices, where their factors sit in the workspace, and the workspace itself
        deferred->mats.clear();
        for (int i = 0; i < nm; ++i) deferred->mats.push_back(SymEigDeferred::Mat{M[i], ws[i].VT, ws[i].Vc, ws[i].TV, ws[i].Q1, ws[i].U, ws[i].W, ws[i].nblk, depth[i]});
        deferred->dbuf.release(); deferred->ibuf.release();
        std::swap(deferred->dbuf.p, dbuf.p); std::swap(deferred->dbuf.bytes, dbuf.bytes);
        std::swap(deferred->ibuf.p, ibuf.p); std::swap(deferred->ibuf.bytes, ibuf.bytes);
        deferred->pending = true;
    }
    hmark("all queued");
    if (host_trace) {
        fprintf(stderr, "[symeig host] %d matrices, nmax %d:", nm, nmax);
        for (const auto& hm : hmarks) fprintf(stderr, "  %s %.0f us", hm.first, hm.second);
        fprintf(stderr, "\n");
    }
    return DMRGX_OK;
}

// Second phase of a deferred call: the root merge's GEMM X[:, kept] = blockdiag(Q_a, Q_b) U[:, kept] and the blocked back-transformation
// X[:, kept] <- H_0 ... H_{n-3} X[:, kept], for the keep[i] largest eigenvalues (the last columns) of every matrix -- half of the flops of
// both, and GEMM launches half as wide, when a truncation keeps half of the states.
dmrgx_status symeig_finish(SymEigDeferred& d, const std::vector<int32_t>& keep, hipStream_t st)
{
    if (!d.pending) return DMRGX_OK;
    const int nm = (int)d.mats.size();
    if ((int)keep.size() != nm) DMRGX_FAIL(DMRGX_ERR_ARG, "symeig_finish: %d counts for %d matrices", (int)keep.size(), nm);
    double* B = d.dbuf.as<double>();
    std::vector<GProd> prods;
    std::vector<GGroup> groups;
    std::vector<GTile> tiles;
    auto add_set = [&](std::vector<GTile>& big, std::vector<GTile>& small) {
        GemmSet s;
        ggemm_schedule(big, groups, 2); ggemm_schedule(small, groups);
        s.big_off = tiles.size(); s.nbig = (int32_t)big.size(); tiles.insert(tiles.end(), big.begin(), big.end());
        s.small_off = tiles.size(); s.nsmall = (int32_t)small.size(); tiles.insert(tiles.end(), small.begin(), small.end());
        return s;
    };
    auto add_gemm = [&](std::vector<GTile>& big, std::vector<GTile>& small, double* C, int ldc, int Mr, int Nc, const double* A, int lda, const double* Bm, int ldb, int K, int accumulate) {
        if (Mr <= 0 || Nc <= 0 || K <= 0) return;
        prods.push_back(GProd{A, Bm, lda, ldb, K, GPROD_GEMM, 1.0});
        groups.push_back(GGroup{C, ldc, Mr, Nc, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, accumulate});
        ggemm_append_tiles_mixed(big, small, (int32_t)groups.size() - 1, Mr, Nc, (K + GG_BK - 1) / GG_BK);
    };
    int max_nblk = 0;
    GemmSet root;
    {
        std::vector<GTile> gb, gs;
        for (int i = 0; i < nm; ++i) {
            const SymEigDeferred::Mat& q = d.mats[(size_t)i];
            const int n = q.m.n, c = keep[(size_t)i];
            if (c < 0 || c > n) DMRGX_FAIL(DMRGX_ERR_ARG, "symeig_finish: matrix %d of order %d cannot keep %d eigenvectors", i, n, c);
            if (c > 0) max_nblk = std::max(max_nblk, q.nblk);
            if (q.depth < 1 || c == 0) continue;                  // (a single leaf: its eigenvectors are complete)
            const int mid = tree_bounds(n, 1)[1], c0 = n - c;
            double* X = q.m.X; const int ldx = q.m.ldx;
            const double* Qs = B + q.Q1; const double* U = B + q.U;        // the root's children live in Q[1] (ld n), its merge matrix in U (ld n)
            add_gemm(gb, gs, X + c0, ldx, mid, c, Qs, n, U + c0, n, mid, 0);
            add_gemm(gb, gs, X + (int64_t)mid * ldx + c0, ldx, n - mid, c, Qs + (int64_t)mid * n + mid, n, U + (int64_t)mid * n + c0, n, n - mid, 0);
        }
        root = add_set(gb, gs);
    }
    std::vector<GemmSet> bt_w((size_t)max_nblk), bt_x((size_t)max_nblk);
    for (int s = 0; s < max_nblk; ++s) {
        std::vector<GTile> wb, wsm, xb, xs;
        for (int i = 0; i < nm; ++i) {
            const SymEigDeferred::Mat& q = d.mats[(size_t)i];
            const int b = q.nblk - 1 - s, n = q.m.n, c = keep[(size_t)i];
            if (b < 0 || c == 0) continue;
            const int b0 = b * WY_NB, kb = std::min(WY_NB, n - 2 - b0), r0 = b0 + 1, c0 = n - c;
            double* X = q.m.X; const int ldx = q.m.ldx;
            add_gemm(wb, wsm, B + q.W, n, kb, c, B + q.TV + (int64_t)b0 * n + r0, n, X + (int64_t)r0 * ldx + c0, ldx, n - r0, 0);      // W = -(T V^T) X[:, kept]
            add_gemm(xb, xs, X + (int64_t)r0 * ldx + c0, ldx, n - r0, c, B + q.Vc + (int64_t)r0 * n + b0, n, B + q.W, n, kb, 1);       // X[:, kept] += V W
        }
        bt_w[(size_t)s] = add_set(wb, wsm);
        bt_x[(size_t)s] = add_set(xb, xs);
    }
    if (!tiles.empty()) {
        DevBuf d_tab;
        PackedUpload pk;
        const size_t o_prods = pk.add(prods), o_groups = pk.add(groups), o_tiles = pk.add(tiles);
        DMRGX_CHK(pk.upload(d_tab, st));
        const GProd* d_prods = packed_at<GProd>(d_tab, o_prods);
        const GGroup* d_groups = packed_at<GGroup>(d_tab, o_groups);
        const GTile* d_tiles = packed_at<GTile>(d_tab, o_tiles);
        auto run_set = [&](const GemmSet& s) -> dmrgx_status {
            DMRGX_CHK(ggemm_launch(d_tiles + s.big_off, d_groups, d_prods, s.nbig, st, 1));
            DMRGX_CHK(ggemm_launch(d_tiles + s.small_off, d_groups, d_prods, s.nsmall, st, 0));
            return DMRGX_OK;
        };
        DMRGX_CHK(run_set(root));
        for (int s = 0; s < max_nblk; ++s) { DMRGX_CHK(run_set(bt_w[(size_t)s])); DMRGX_CHK(run_set(bt_x[(size_t)s])); }
    }
    // (the tables and the workspace go back to the stream-ordered pool: later users are ordered behind the launches above)
    d.dbuf.release(); d.ibuf.release();
    d.pending = false;
    return DMRGX_OK;
}

}  // namespace dmrgx
