// K2: lowest eigenpair of the planned superblock Hamiltonian -- thick-restart Lanczos (== Krylov-Schur for a
// symmetric operator) with classical Gram-Schmidt + one refinement pass (CGS2), everything device-resident.
//
// Replaces the SLEPc solve configured at reference include/DMRGBlockContainer.hpp:1488-1499 (EPS_HEP,
// EPS_SMALLEST_REAL, nev = 1; SLEPc's defaults for that call: Krylov-Schur, ncv = 16, relative residual
// tol = 1e-8 -- third-party, not in the reference tree).  One "superblock MatMult" == one dmrgx_kron_apply.
//
// Per Lanczos step the vector work is three fused HBM-bound passes over the basis instead of BLAS-1 calls (CGS2):
//   multi_dot       : c = V^T w                                  (reads j+2 vectors)
//   axpy_dot        : w' = w - V c ; c2 = V^T w' , w'.w'         (reads j+2 vectors, writes w')
//   axpy_normalise  : v_{j+1} = (w' - V c2) / sqrt(w'.w' - |c2|^2)   (reads j+2 vectors, writes v_{j+1})  Nothing is copied to the host inside a restart cycle: the normalisation reads beta^2
// from device memory, and the host fetches the projected matrix once per cycle (ncv/2 MatMults).
// world_size > 1: vectors are this rank's stripe segment; per Lanczos step there are exactly two fused all-reduces (of
// j+2 doubles each) and one all-gather of the Krylov vector before the MatMult.
#include "common.h"
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstddef>

namespace dmrgx {
namespace {

constexpr int DOT_BLOCKS = 1024, DOT_THREADS = 256, DOT_CHUNK = 8, MAX_NCV = 64, FUSE_NV = 24, GD_NV = 8;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// partial[(chunk*DOT_CHUNK + i) * DOT_BLOCKS + block] = sum_e V[(v0+i)*ldv + e] * w[e]   (i < DOT_CHUNK)
// the last chunk's slot `nv` holds w.w
template <bool VEC2>                     // VEC2: n even and 16-byte aligned rows -> one 16-byte load per lane and vector
__global__ void __launch_bounds__(DOT_THREADS)
multi_dot_kernel(const double* __restrict__ V, int64_t ldv, int nv, const double* __restrict__ w, int64_t n, double* __restrict__ partial)
{
    __shared__ double red[DOT_THREADS / 64][DOT_CHUNK];
    const int v0 = blockIdx.y * DOT_CHUNK;
    double acc[DOT_CHUNK];
#pragma unroll
    for (int i = 0; i < DOT_CHUNK; ++i) acc[i] = 0.0;
    if (VEC2) {
        const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
        for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n / 2; e += (int64_t)gridDim.x * DOT_THREADS) {
            const double2 wv = w2[e];
#pragma unroll
            for (int i = 0; i < DOT_CHUNK; ++i) {
                const int v = v0 + i;
                if (v < nv) { const double2 x = reinterpret_cast<const double2*>(V + (int64_t)v * ldv)[e]; acc[i] += x.x * wv.x + x.y * wv.y; }
                else if (v == nv) acc[i] += wv.x * wv.x + wv.y * wv.y;
            }
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * DOT_THREADS) {
            const double wv = w[e];
#pragma unroll
            for (int i = 0; i < DOT_CHUNK; ++i) {
                const int v = v0 + i;
                if (v < nv) acc[i] += V[(int64_t)v * ldv + e] * wv;
                else if (v == nv) acc[i] += wv * wv;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < DOT_CHUNK; ++i) {
        const double s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < DOT_CHUNK) {
        double s = 0.0;
        for (int wv = 0; wv < DOT_THREADS / 64; ++wv) s += red[wv][threadIdx.x];
        partial[(int64_t)(v0 + threadIdx.x) * gridDim.x + blockIdx.x] = s;
    }
}

// out[i] = sum_b partial[i*DOT_BLOCKS + b]   (deterministic order)
__global__ void __launch_bounds__(DOT_THREADS)
reduce_partials_kernel(const double* __restrict__ partial, double* __restrict__ out, int count, int nblk)
{
    __shared__ double red[DOT_THREADS / 64];
    const int i = blockIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += DOT_THREADS) s += partial[(int64_t)i * nblk + b];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < DOT_THREADS / 64; ++k) t += red[k]; out[i] = t; }
}

// w -= sum_i c[i] V[i] ; partial[b] = sum_e w[e]^2 ; hacc[i] += c[i] (accumulated projection coefficients)
__global__ void __launch_bounds__(DOT_THREADS)
multi_axpy_kernel(const double* __restrict__ V, int64_t ldv, int nv, const double* __restrict__ c, double* __restrict__ w, int64_t n,
                  double* __restrict__ partial, double* __restrict__ hacc)
{
    __shared__ double cs[MAX_NCV + 1];
    __shared__ double red[DOT_THREADS / 64];
    if (threadIdx.x < nv) cs[threadIdx.x] = c[threadIdx.x];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < nv && hacc) hacc[threadIdx.x] += cs[threadIdx.x];
    double nrm = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * DOT_THREADS) {
        double x = w[e];
        for (int i = 0; i < nv; ++i) x -= cs[i] * V[(int64_t)i * ldv + e];
        w[e] = x;
        nrm += x * x;
    }
    nrm = wave_sum(nrm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < DOT_THREADS / 64; ++k) t += red[k]; partial[blockIdx.x] = t; }
}

// First Gram-Schmidt pass fused with the dots of the second: w' = w - V c is formed per element and immediately
// multiplied into this thread's partial sums of V^T w' and w'.w' (V's values are still in registers), so the basis is
// read from HBM once for both.  nv <= NVMAX: FUSE_NV for the Lanczos path (the default ncv = 16 gives nv <= 17), GD_NV for the
// small search space of the Davidson path (a third of the registers).
// partial[i * DOT_BLOCKS + block], i < nv: V_i . w' ; i == nv: w'.w'
template <int NVMAX>
__global__ void __launch_bounds__(DOT_THREADS)
axpy_dot_kernel(const double* __restrict__ V, int64_t ldv, int nv, const double* __restrict__ c, double* __restrict__ w, int64_t n,
                double* __restrict__ partial, double* __restrict__ hacc)
{
    __shared__ double cs[NVMAX];
    __shared__ double red[DOT_THREADS / 64][NVMAX + 1];
    if (threadIdx.x < nv) cs[threadIdx.x] = c[threadIdx.x];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < nv && hacc) hacc[threadIdx.x] += cs[threadIdx.x];
    double acc[NVMAX + 1];
#pragma unroll
    for (int i = 0; i <= NVMAX; ++i) acc[i] = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * DOT_THREADS) {
        double v[NVMAX];
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) v[i] = i < nv ? V[(int64_t)i * ldv + e] : 0.0;
        double x = w[e];
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) if (i < nv) x -= cs[i] * v[i];
        w[e] = x;
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) if (i < nv) acc[i] += v[i] * x;
        acc[NVMAX] += x * x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i <= NVMAX; ++i) {
        if (i < nv || i == NVMAX) {
            const double s2 = wave_sum(acc[i]);
            if (lane == 0) red[wave][i] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x <= nv) {
        const int src = threadIdx.x < nv ? threadIdx.x : NVMAX;
        double s2 = 0.0;
        for (int wv = 0; wv < DOT_THREADS / 64; ++wv) s2 += red[wv][src];
        partial[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s2;
    }
}

// Second Gram-Schmidt pass fused with the normalisation: dst = (w - V c) / beta with beta^2 = ||w - V c||^2 = c[nv] - |c|^2
// (c[nv] = w.w; V orthonormal and c = V^T w the tiny refinement coefficients, so there is no cancellation): every
// workgroup recomputes beta^2 from the nv+1 reduced dots -- no separate reduction pass, no extra all-reduce and no extra
// launch -- and workgroup 0 records it in beta2_out (the step's row of the projected matrix).
template <bool VEC2>
__global__ void __launch_bounds__(DOT_THREADS)
axpy_normalise_kernel(const double* __restrict__ V, int64_t ldv, int nv, const double* __restrict__ c, const double* __restrict__ w,
                      double* __restrict__ dst, int64_t n, double* __restrict__ beta2_out, double* __restrict__ hacc)
{
    __shared__ double cs[MAX_NCV + 2];
    if (threadIdx.x <= nv) cs[threadIdx.x] = c[threadIdx.x];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < nv && hacc) hacc[threadIdx.x] += cs[threadIdx.x];
    double s2 = cs[nv], sc = 0.0;
    for (int i = 0; i < nv; ++i) { s2 -= cs[i] * cs[i]; sc += cs[i] * cs[i]; }          // same order in every thread: one value for the whole grid
    s2 = s2 > 0.0 ? s2 : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && beta2_out) *beta2_out = s2;
    const double inv = s2 > 1e-290 ? 1.0 / sqrt(s2) : 0.0;
    // The second Gram-Schmidt pass is a correction of relative size |c| / |w|: when the first pass already left less than 3e-14 of
    // the vector inside the basis (the usual case: ~1e-15) the basis is not read again -- the vector is only scaled.  (Grid-uniform.
    // 1e-12 was too loose: the step-by-step correlator parity of the 4x4 case, asked at 1e-12 absolute, moved by 1.2e-12.)
    if (sc <= 1e-27 * cs[nv]) nv = 0;
    if (VEC2) {
        const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
        double2* __restrict__ d2 = reinterpret_cast<double2*>(dst);
        for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n / 2; e += (int64_t)gridDim.x * DOT_THREADS) {
            double2 x = w2[e];
            for (int i = 0; i < nv; ++i) { const double2 v = reinterpret_cast<const double2*>(V + (int64_t)i * ldv)[e]; x.x -= cs[i] * v.x; x.y -= cs[i] * v.y; }
            x.x *= inv; x.y *= inv;
            d2[e] = x;
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * DOT_THREADS) {
            double x = w[e];
            for (int i = 0; i < nv; ++i) x -= cs[i] * V[(int64_t)i * ldv + e];
            dst[e] = x * inv;
        }
    }
}

// ||w - V c||^2 = w.w - |c|^2 for orthonormal V and c = V^T w (c is the tiny refinement coefficient vector of the second
// Gram-Schmidt pass, so there is no cancellation): saves a reduction pass and, distributed, one all-reduce per step.
__global__ void norm_after_projection_kernel(const double* __restrict__ c, int nv, double* __restrict__ nrm2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { double s = c[nv]; for (int i = 0; i < nv; ++i) s -= c[i] * c[i]; *nrm2 = s > 0.0 ? s : 0.0; }
}

// dst = src * (nrm2 > tiny ? 1/sqrt(nrm2) : 0)
__global__ void scale_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n, const double* __restrict__ nrm2)
{
    const double s2 = *nrm2;
    const double inv = s2 > 1e-290 ? 1.0 / sqrt(s2) : 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) dst[e] = src[e] * inv;
}

// out[j][e] = sum_i V[i][e] Q[i*ldq + j0 + j]   for j < min(DOT_CHUNK, k - j0), j0 = blockIdx.y*DOT_CHUNK
__global__ void __launch_bounds__(DOT_THREADS)
basis_rotate_kernel(const double* __restrict__ V, int64_t ldv, int m, const double* __restrict__ Q, int ldq, int k,
                    double* __restrict__ out, int64_t ldo, int64_t n)
{
    __shared__ double qs[MAX_NCV * DOT_CHUNK];
    const int j0 = blockIdx.y * DOT_CHUNK;
    for (int t = threadIdx.x; t < m * DOT_CHUNK; t += DOT_THREADS) {
        const int i = t / DOT_CHUNK, j = t % DOT_CHUNK;
        qs[t] = (j0 + j < k) ? Q[i * ldq + j0 + j] : 0.0;
    }
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n; e += (int64_t)gridDim.x * DOT_THREADS) {
        double acc[DOT_CHUNK];
#pragma unroll
        for (int j = 0; j < DOT_CHUNK; ++j) acc[j] = 0.0;
        for (int i = 0; i < m; ++i) {
            const double v = V[(int64_t)i * ldv + e];
#pragma unroll
            for (int j = 0; j < DOT_CHUNK; ++j) acc[j] += v * qs[i * DOT_CHUNK + j];
        }
#pragma unroll
        for (int j = 0; j < DOT_CHUNK; ++j) if (j0 + j < k) out[(int64_t)(j0 + j) * ldo + e] = acc[j];
    }
}

// ---- generalized Davidson: the small problem lives on the device ------------------------------------------------------------------
// State of the projected problem, one device block of doubles (GdState offsets): nothing of it is needed on the host, so an iteration
// has no read-back / host solve / upload in its dependency chain (rounds 2-4: one blocking synchronisation per MatMult, 0.83 ms of idle
// GPU per configs[3] step, 0.19 ms of a 3.2 ms step at m = 512).  The host only LOOKS at (|r|^2, theta) through pinned memory, one
// iteration behind the queue.
constexpr int GD_MAX = FUSE_NV;              // largest search space (ncv) the Davidson path accepts
constexpr int GD_LD = GD_MAX;                // row stride of the small matrices
struct GdState {
    double G[GD_MAX * GD_LD];                // projected matrix V^T H V (symmetric, valid in [0, mm) x [0, mm))
    double Y[GD_MAX * GD_LD];                // Ritz basis of the last solve (columns), ascending Ritz values
    double th[GD_MAX];                       // Ritz values
    double y[GD_MAX];                        // coefficients of the current Ritz vector in the current basis
    double yprev[GD_MAX];                    // ... of the previous iteration's Ritz vector (GD+k restart)
    double Q[GD_MAX * GD_LD];                // restart rotation, row-major mm x kk with ld = kk
    double theta;                            // lowest Ritz value
    double hv0_2;                            // |H v_0|^2 (start-vector check: not positive <=> zero / NaN start vector)
    double e0[GD_MAX];                       // (1, 0, 0, ...): the Ritz vector's coefficients right after a restart
    int after_restart;                       // the basis was rotated since y was computed: the previous Ritz vector is the first basis vector
    int pad_;
};

// One workgroup: the newest column of G (reduced here from the per-block partial sums of multi_dot_kernel, or taken reduced when an
// all-reduce had to come first), the symmetric eigenproblem of G by cyclic Jacobi with round-robin pair ordering (all n/2 rotations of a
// round at once), warm-started in the basis [previous Ritz vectors | new direction] where G is an arrowhead; at a restart the rotation
// Q = [keep lowest Ritz vectors | previous iteration's Ritz vector, orthonormalised] and G <- Q^T G Q.
constexpr int RITZ_THREADS = 64;
__global__ void __launch_bounds__(RITZ_THREADS)
gd_ritz_kernel(GdState* __restrict__ S, const double* __restrict__ col_src, int col_is_partial, int nblk, int mm, int warm, int restart, int keep)
{
    __shared__ double A[GD_MAX][GD_MAX + 1], Qa[GD_MAX][GD_MAX + 1], Yn[GD_MAX][GD_MAX + 1], Yp[GD_MAX][GD_MAX + 1];
    __shared__ double col[GD_MAX + 1], cs_c[GD_MAX / 2], cs_s[GD_MAX / 2], dg[GD_MAX], thp[GD_MAX], ths[GD_MAX];
    __shared__ int pp[GD_MAX / 2], qq[GD_MAX / 2], perm[GD_MAX];
    __shared__ double offn, dgn;
    const int tid = threadIdx.x, lane = tid;                                // ONE wave: every barrier below is a wait on the wave's own LDS traffic
    const int j = mm - 1;
    // the device state that the loops below read goes to LDS in one pass, beside the partial sums' loads (a global load inside a serial
    // loop of one wave is a microsecond per trip: the first build of this kernel took 47-67 us, most of it such loops)
    if (warm && mm > 1) {
        for (int e = tid; e < (mm - 1) * (mm - 1); e += RITZ_THREADS) Yp[e / (mm - 1)][e % (mm - 1)] = S->Y[(e / (mm - 1)) * GD_LD + e % (mm - 1)];
        if (tid < mm - 1) thp[tid] = S->th[tid];
    }
    const double yold = tid < GD_MAX ? S->y[tid] : 0.0;
    const int was_restart = S->after_restart;
    // ---- newest column of G: sums of the per-block partials in fixed order (all mm + 1 sums at once: one load latency) ---------------
    if (col_is_partial) {
        // every load of a group of four sums is issued before the first is used (16 per lane and sum: left to a loop with a carried add the
        // compiler waits for each load in turn -- ~130 dependent round trips to the L2 were 40 of this kernel's 45-60 us)
        constexpr int PER = DOT_BLOCKS / RITZ_THREADS;
        for (int i0 = 0; i0 <= mm; i0 += 4) {
            double t[4][PER];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < PER; ++u) t[g][u] = (i0 + g <= mm && tid + u * RITZ_THREADS < nblk) ? col_src[(int64_t)(i0 + g) * nblk + tid + u * RITZ_THREADS] : 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double v = 0.0;
#pragma unroll
                for (int u = 0; u < PER; ++u) v += t[g][u];
                v = wave_sum(v);
                if (lane == 0 && i0 + g <= mm) col[i0 + g] = v;
            }
        }
    } else if (tid <= mm) col[tid] = col_src[tid];
    __syncthreads();
    if (tid < mm) { S->G[tid * GD_LD + j] = col[tid]; S->G[j * GD_LD + tid] = col[tid]; }
    if (tid == 0 && mm == 1) S->hv0_2 = col[1];
    if (tid < GD_MAX && mm == 1) S->e0[tid] = tid == 0 ? 1.0 : 0.0;
    // the previous Ritz vector in this basis: its coefficients padded with a zero, or -- a restart since -- the first basis vector
    if (tid < mm) S->yprev[tid] = was_restart ? (tid == 0 ? 1.0 : 0.0) : (tid < mm - 1 ? yold : 0.0);
    if (tid == 0) S->after_restart = restart;
    const int n = (mm + 1) & ~1;                                          // even order for the pair schedule (padding: a decoupled zero row)
    // ---- matrix to diagonalise: arrowhead in the previous Ritz basis (warm) or G itself ------------------------------------------
    const int mp = mm - 1;
    for (int e = tid; e < n * n; e += RITZ_THREADS) {
        const int a = e / n, b = e % n;
        double v = 0.0;
        if (a < mm && b < mm) {
            if (!warm || mp == 0) v = (a == j || b == j) ? col[a == j ? b : a] : S->G[a * GD_LD + b];
            else if (a < mp && b < mp) v = a == b ? thp[a] : 0.0;
            else if (a == mp && b == mp) v = col[j];
            else {                                                        // border: b_i = sum_k Yp[k][i] g[k]
                const int i = a == mp ? b : a;
                for (int k = 0; k < mp; ++k) v += Yp[k][i] * col[k];
            }
        }
        A[a][b] = v;
        Qa[a][b] = a == b ? 1.0 : 0.0;
    }
    __syncthreads();
    // ---- cyclic Jacobi, round-robin ordering: round r pairs (r, n-1) and ((r + k) mod (n-1), (r - k) mod (n-1)), k = 1 .. n/2 - 1 --
    // (index arithmetic with a run-time divisor and f64 divisions / square roots are long instruction sequences on this ISA and a single
    //  wave has nothing to hide them behind: the work items of a lane are decoded once, a rotation takes one division and two rsqrt-class
    //  operations, the modulo of the schedule is a conditional subtraction)
    constexpr int UMAX = (GD_MAX / 2 * GD_MAX + RITZ_THREADS - 1) / RITZ_THREADS;
    int wk[UMAX], wi[UMAX];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) { const int e = tid + u * RITZ_THREADS; wk[u] = e < (n / 2) * n ? e / n : -1; wi[u] = e % n; }
    for (int sweep = 0; sweep < 30 && n > 1; ++sweep) {
        {
            double off = 0.0, dgs = 0.0;
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {                                   // (the n/2 x n work items cover the upper half of the rows: both halves here)
                if (wk[u] < 0) continue;
                const double v0 = A[wk[u]][wi[u]], v1 = A[wk[u] + n / 2][wi[u]];
                if (wk[u] == wi[u]) dgs += v0 * v0; else off += v0 * v0;
                if (wk[u] + n / 2 == wi[u]) dgs += v1 * v1; else off += v1 * v1;
            }
            off = wave_sum(off); dgs = wave_sum(dgs);
            if (lane == 0) { offn = off; dgn = dgs; }
        }
        __syncthreads();
        if (offn <= 1e-29 * (dgn + offn) || offn == 0.0) break;               // off-diagonal norm <= 3e-15 ||A|| (n eps ||A|| is the floor of the rotations' own round-off)
        for (int r = 0; r < n - 1; ++r) {
            if (tid < n / 2) {
                const int k = tid;
                int p0 = r + k, q0 = r - k + (n - 1);
                if (p0 >= n - 1) p0 -= n - 1;
                if (q0 >= n - 1) q0 -= n - 1;
                if (k == 0) { p0 = r; q0 = n - 1; }
                const int p = min(p0, q0), q = max(p0, q0);
                const double apq = A[p][q];
                double c = 1.0, sn = 0.0;
                if (apq != 0.0) {
                    const double th = 0.5 * (A[q][q] - A[p][p]);
                    const double h = sqrt(th * th + apq * apq);
                    const double t = apq / (th + (th >= 0.0 ? h : -h));             // the smaller root of t^2 + 2 (th / apq) t - 1 = 0
                    c = rsqrt(1.0 + t * t); sn = t * c;
                }
                pp[k] = p; qq[k] = q; cs_c[k] = c; cs_s[k] = sn;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {                                   // columns p, q of A and of the accumulated rotation
                if (wk[u] < 0) continue;
                const int k = wk[u], i = wi[u];
                const int p = pp[k], q = qq[k];
                const double c = cs_c[k], sn = cs_s[k];
                const double aip = A[i][p], aiq = A[i][q];
                A[i][p] = c * aip - sn * aiq; A[i][q] = sn * aip + c * aiq;
                const double qip = Qa[i][p], qiq = Qa[i][q];
                Qa[i][p] = c * qip - sn * qiq; Qa[i][q] = sn * qip + c * qiq;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {                                   // rows p, q of A
                if (wk[u] < 0) continue;
                const int k = wk[u], i = wi[u];
                const int p = pp[k], q = qq[k];
                const double c = cs_c[k], sn = cs_s[k];
                const double api = A[p][i], aqi = A[q][i];
                // (the rotated pair is annihilated exactly: computed, it is round-off of the size of the stopping rule's threshold and the
                //  sweeps never end -- the first builds of this kernel ran all 30 sweeps of every call, 45-65 us)
                A[p][i] = i == q ? 0.0 : c * api - sn * aqi; A[q][i] = i == p ? 0.0 : sn * api + c * aqi;
            }
            __syncthreads();
        }
    }
    // ---- ascending order (rank by counting; the padding row, if any, is dropped) ------------------------------------------------
    if (tid < mm) dg[tid] = A[tid][tid];
    __syncthreads();
    if (tid < mm) {
        int rank = 0;
        for (int o = 0; o < mm; ++o) rank += (dg[o] < dg[tid] || (dg[o] == dg[tid] && o < tid)) ? 1 : 0;
        perm[rank] = tid;
    }
    __syncthreads();
    // Y = blkdiag(Yp, 1) Qa (warm) or Qa, columns in ascending order
    for (int e = tid; e < mm * mm; e += RITZ_THREADS) {
        const int i = e / mm, cidx = e % mm;
        const int src = perm[cidx];
        double v;
        if (!warm || mp == 0 || i == mp) v = Qa[i][src];
        else { v = 0.0; for (int k = 0; k < mp; ++k) v += Yp[i][k] * Qa[k][src]; }
        Yn[i][cidx] = v;
    }
    __syncthreads();
    for (int e = tid; e < mm * mm; e += RITZ_THREADS) S->Y[(e / mm) * GD_LD + (e % mm)] = Yn[e / mm][e % mm];
    if (tid < mm) { ths[tid] = dg[perm[tid]]; S->th[tid] = ths[tid]; S->y[tid] = Yn[tid][0]; }
    if (tid == 0) S->theta = dg[perm[0]];
    if (!restart) return;
    // ---- restart: Q = [Ritz vectors 0 .. keep-1 | previous Ritz vector orthonormalised against them] (keep + 1 columns) -----------
    __syncthreads();
    const int kk = keep + 1;
    {
        // p = yprev - sum_c Q_c (Q_c . yprev), twice; lanes hold the entries
        double pv = lane < mm ? S->yprev[lane] : 0.0;
        for (int pass = 0; pass < 2; ++pass)
            for (int c = 0; c < keep; ++c) {
                const double qc = lane < mm ? Yn[lane][c] : 0.0;
                double d = qc * pv;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
                pv -= d * qc;
            }
        double nn = pv * pv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
        // (a previous Ritz vector that lies in the span of the kept ones adds nothing: the next Ritz vector takes its place)
        const double last = nn > 1e-16 ? pv / sqrt(nn) : (lane < mm ? Yn[lane][keep] : 0.0);
        if (lane < mm) A[lane][keep] = last;                              // A is free: it now holds Q (mm x kk)
    }
    for (int e = tid; e < mm * keep; e += RITZ_THREADS) A[e / keep][e % keep] = Yn[e / keep][e % keep];
    __syncthreads();
    for (int e = tid; e < mm * kk; e += RITZ_THREADS) S->Q[e] = A[e / kk][e % kk];
    // G <- Q^T G Q: diag(theta) on the Ritz block exactly, the border through G q
    for (int e = tid; e < mm * mm; e += RITZ_THREADS) Qa[e / mm][e % mm] = S->G[(e / mm) * GD_LD + e % mm];      // (Qa is free: G in LDS)
    __syncthreads();
    if (tid < mm) { double v = 0.0; for (int b = 0; b < mm; ++b) v += Qa[tid][b] * A[b][keep]; col[tid] = v; }      // G q_last
    __syncthreads();
    if (tid < kk) { double v = 0.0; for (int a = 0; a < mm; ++a) v += A[a][tid] * col[a]; dg[tid] = v; }                      // Q^T G q_last
    __syncthreads();
    for (int e = tid; e < GD_MAX * GD_LD; e += RITZ_THREADS) {
        const int a = e / GD_LD, b = e % GD_LD;
        double v = 0.0;
        if (a < keep && b < keep) v = a == b ? ths[a] : 0.0;
        else if (a < kk && b < kk) v = dg[a == keep ? b : a];
        S->G[e] = v;
    }
    // (y stays: the correction kernel that follows still works in the old basis; after_restart tells the next call)
}

// u = V y, r = W y - theta u (W = H V), t = r / (theta - D) with |theta - D| kept away from 0; y and theta are read from the device
// state.  Writes t; partial: [i * nblk + b], i < nv: V_i . t (the first Gram-Schmidt pass of the correction, while V's values are in
// registers: nv <= NVMAX = 8 values per lane -- with the 17 of rounds 3-4's search space this fusion cost a wave per SIMD and was
// dropped, DESIGN K2); i == nv: t . t; i == nv + 1: r . r.  DOTS = false: only r . r, at index 0.
template <int NVMAX, bool DOTS>
__global__ void __launch_bounds__(DOT_THREADS)
ritz_precond_kernel(const double* __restrict__ V, const double* __restrict__ W, int64_t ldv, int nv, const GdState* __restrict__ S,
                    const double* __restrict__ D, double floor_rel, double* __restrict__ t, int64_t n, double* __restrict__ partial)
{
    // (Olsen's correction t - (u.t / u.t2) t2 with t2 = u / (theta - D) was measured in round 3 -- 2 436 instead of 2 431 MatMults in a
    //  sweep of configs[3]: with a diagonal preconditioner it is noise -- and is not part of the library)
    __shared__ double ys[NVMAX];
    __shared__ double red[DOT_THREADS / 64][NVMAX + 2];
    if (threadIdx.x < NVMAX) ys[threadIdx.x] = threadIdx.x < nv ? S->y[threadIdx.x] : 0.0;
    const double theta = S->theta;
    const double floor_ = floor_rel * fmax(1.0, fabs(theta)) * 1e-2 + 1e-12;
    __syncthreads();
    double acc[NVMAX + 2];
#pragma unroll
    for (int i = 0; i < NVMAX + 2; ++i) acc[i] = 0.0;
    // two elements per lane and 16-byte loads: n and ldv are even, every row 16-byte aligned (the caller pads its vectors, eigs_davidson)
    const double2* __restrict__ D2 = reinterpret_cast<const double2*>(D);
    double2* __restrict__ t2 = reinterpret_cast<double2*>(t);
    for (int64_t e = (int64_t)blockIdx.x * DOT_THREADS + threadIdx.x; e < n / 2; e += (int64_t)gridDim.x * DOT_THREADS) {
        double2 v[NVMAX];
        double2 u = {0.0, 0.0}, hu = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NVMAX; ++i)
            if (i < nv) {
                v[i] = reinterpret_cast<const double2*>(V + (int64_t)i * ldv)[e];
                const double2 w = reinterpret_cast<const double2*>(W + (int64_t)i * ldv)[e];
                u.x += ys[i] * v[i].x; u.y += ys[i] * v[i].y; hu.x += ys[i] * w.x; hu.y += ys[i] * w.y;
            } else v[i] = double2{0.0, 0.0};
        const double2 r = {hu.x - theta * u.x, hu.y - theta * u.y};
        const double2 dv = D2[e];
        double2 den = {theta - dv.x, theta - dv.y};
        if (fabs(den.x) < floor_) den.x = den.x < 0.0 ? -floor_ : floor_;
        if (fabs(den.y) < floor_) den.y = den.y < 0.0 ? -floor_ : floor_;
        const double2 te = {r.x / den.x, r.y / den.y};
        t2[e] = te;
        if (DOTS) {
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv) acc[i] += v[i].x * te.x + v[i].y * te.y;
            acc[NVMAX] += te.x * te.x + te.y * te.y;
        }
        acc[NVMAX + 1] += r.x * r.x + r.y * r.y;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NVMAX + 2; ++i) {
        if ((DOTS && (i < nv || i == NVMAX)) || i == NVMAX + 1) {
            const double s2 = wave_sum(acc[i]);
            if (lane == 0) red[wave][i] = s2;
        }
    }
    __syncthreads();
    if (DOTS) {
        if (threadIdx.x <= nv + 1) {
            const int src = threadIdx.x < nv ? threadIdx.x : (threadIdx.x == nv ? NVMAX : NVMAX + 1);
            double s2 = 0.0;
            for (int wv = 0; wv < DOT_THREADS / 64; ++wv) s2 += red[wv][src];
            partial[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = s2;
        }
    } else if (threadIdx.x == 0) {
        double s2 = 0.0;
        for (int wv = 0; wv < DOT_THREADS / 64; ++wv) s2 += red[wv][NVMAX + 1];
        partial[blockIdx.x] = s2;
    }
}

// zero pads of the Davidson work vectors (eigs_davidson): W[j * ld + n] for j < count, t[n], D[n]
__global__ void gd_zero_pads_kernel(double* __restrict__ W, int64_t ld, int count, int64_t n, double* __restrict__ t, double* __restrict__ D)
{
    for (int j = threadIdx.x; j < count; j += blockDim.x) W[(int64_t)j * ld + n] = 0.0;
    if (threadIdx.x == 0) { t[n] = 0.0; D[n] = 0.0; }
}

// out[i] = sum_b partial[i * nblk + b] (as reduce_partials_kernel); the block of index `look_idx` also hands (|r|^2, theta, |H v_0|^2) to
// the host's look buffer (pinned memory, fetched behind an event).  look_only: the sums are in `out` already (an all-reduce came between).
__global__ void __launch_bounds__(DOT_THREADS)
gd_reduce_look_kernel(const double* __restrict__ partial, double* __restrict__ out, int count, int nblk, int look_idx, int look_only,
                      const GdState* __restrict__ S, double* __restrict__ look)
{
    __shared__ double red[DOT_THREADS / 64];
    const int i = blockIdx.x;
    if (i >= count) return;
    double t = 0.0;
    if (!look_only) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += DOT_THREADS) s += partial[(int64_t)i * nblk + b];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) { for (int k = 0; k < DOT_THREADS / 64; ++k) t += red[k]; out[i] = t; }
    } else if (threadIdx.x == 0) t = out[i];
    if (threadIdx.x == 0 && i == look_idx) { look[0] = t; look[1] = S->theta; look[2] = S->hv0_2; }
}

// counter-based uniform(-1,1) start vector (splitmix64 of seed + global index)
__global__ void random_fill_kernel(double* __restrict__ v, int64_t n, uint64_t seed, int64_t index_offset)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(e + index_offset + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        v[e] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

// symmetric eigen-decomposition of a small dense matrix by cyclic Jacobi (host); ascending eigenvalues,
// eigenvectors in the columns of Q (row-major n x n)
void jacobi_eigh(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& Q)
{
    Q.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq; A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk; A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double qkp = Q[(size_t)k * n + p], qkq = Q[(size_t)k * n + q];
                    Q[(size_t)k * n + p] = c * qkp - s * qkq; Q[(size_t)k * n + q] = s * qkp + c * qkq;
                }
            }
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A[(size_t)a * n + a] < A[(size_t)b * n + b]; });
    w.resize(n);
    std::vector<double> Qs((size_t)n * n);
    for (int j = 0; j < n; ++j) { w[j] = A[(size_t)idx[j] * n + idx[j]]; for (int i = 0; i < n; ++i) Qs[(size_t)i * n + j] = Q[(size_t)i * n + idx[j]]; }
    Q.swap(Qs);
}

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

static dmrgx_status eigs_davidson(dmrgx_kron_plan* plan, const dmrgx_eigs_opts* opts, double* e0, double* psi_full, dmrgx_eigs_stats* stats, hipStream_t st);

namespace {
// Where a distributed MatMult spends its time, for the N > 1 bench line: three events per MatMult on the solver's
// stream, read when the timer goes out of scope (the solve has synchronised by then), summed into process-wide totals.
std::mutex g_ct_mutex;
double g_ct_allgather_ms = 0.0, g_ct_apply_ms = 0.0;
int64_t g_ct_matvecs = 0;
struct CommTimer {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    explicit CommTimer(hipStream_t s) : st(s) {}
    void mark()
    {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev.push_back(e); }
        (void)hipEventRecord(ev[used++], st);
    }
    ~CommTimer()
    {
        double ag = 0.0, ap = 0.0; int64_t nmv = 0;
        for (size_t i = 0; i + 2 < used; i += 3) {
            float a = 0.f, b = 0.f;
            if (hipEventSynchronize(ev[i + 2]) != hipSuccess) break;
            if (hipEventElapsedTime(&a, ev[i], ev[i + 1]) == hipSuccess && hipEventElapsedTime(&b, ev[i + 1], ev[i + 2]) == hipSuccess) { ag += a; ap += b; ++nmv; }
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (nmv) { std::lock_guard<std::mutex> lk(g_ct_mutex); g_ct_allgather_ms += ag; g_ct_apply_ms += ap; g_ct_matvecs += nmv; }
    }
};
}  // namespace

extern "C" dmrgx_status dmrgx_eigs_comm_timing(double* allgather_ms, double* apply_ms, int64_t* n_matvec, int32_t reset)
{
    std::lock_guard<std::mutex> lk(g_ct_mutex);
    if (allgather_ms) *allgather_ms = g_ct_allgather_ms;
    if (apply_ms) *apply_ms = g_ct_apply_ms;
    if (n_matvec) *n_matvec = g_ct_matvecs;
    if (reset) { g_ct_allgather_ms = 0.0; g_ct_apply_ms = 0.0; g_ct_matvecs = 0; }
    return DMRGX_OK;
}

// Pinned doubles per host thread for what travels to the host behind the solver's own synchronisations: [0, 8) scalars ([2]: squared
// norm of a caller-supplied start vector), [8, 8 + MAX_NCV + 2) the newest column of the projected matrix of the Davidson iteration
// (a read-back into pageable memory is staged and blocks the host for ~20 us; one per MatMult).
static double* pinned_scalars()
{
    static thread_local double* p = nullptr;
    if (!p && hipHostMalloc((void**)&p, (8 + MAX_NCV + 2) * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    return p;
}
// A start vector is not trusted when its norm is zero / NaN (seen from the first coefficients) or below what the caller asked for.
static bool start_too_light(const dmrgx_eigs_opts* opts, const double* pin)
{
    return opts->min_initial_norm2 > 0.0 && pin && !(pin[2] >= opts->min_initial_norm2);
}

extern "C" dmrgx_status dmrgx_eigs_lowest(dmrgx_kron_plan* plan, const dmrgx_eigs_opts* opts, double* e0,
                                          double* psi_full, dmrgx_eigs_stats* stats, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!plan || !opts || !e0 || !psi_full) DMRGX_FAIL(DMRGX_ERR_ARG, "eigs_lowest: null argument");
    // the diagonally preconditioned iteration pays off from a start vector close to the answer (the engine's transformed ground
    // state: 16-19 MatMults where Lanczos needs 21-27); from a random vector it is twice as slow as Lanczos (measured 140 vs 72),
    // so without a supplied start vector the request falls through to the Lanczos path
    dmrgx_kron_info I;
    DMRGX_CHK(dmrgx_kron_plan_info(plan, &I));
    // (a search space of fewer than two vectors -- a one-state superblock sector, ncv = 1 -- has nothing to precondition: Lanczos path)
    const bool gd_space = std::min<int64_t>(opts->ncv > 0 ? opts->ncv : 16, I.n_states) >= 2;
    if (opts->method == 1 && opts->use_initial && opts->max_matvec <= 0 && gd_space) return eigs_davidson(plan, opts, e0, psi_full, stats, st);
    const bool dist = I.vec_len != I.n_states;
    const bool hooks = opts->allgather && opts->allreduce_sum;
    if (dist && !hooks && !opts->comm) DMRGX_FAIL(DMRGX_ERR_ARG, "eigs_lowest: a striped plan needs a communicator (opts->comm) or the allgather/allreduce hooks");
    auto gather_full = [&](double* full) -> dmrgx_status {      // in-place all-gather of the rank segments of a full vector
        return hooks ? opts->allgather(opts->user, full, I.seg_stride, st) : dmrgx_comm_allgather(opts->comm, full, I.seg_stride, st);
    };
    const int64_t n = I.local_len, N = I.n_states;
    int m = opts->ncv > 0 ? opts->ncv : 16;
    m = (int)std::min<int64_t>(std::min(m, MAX_NCV), N);
    if (m < 1) DMRGX_FAIL(DMRGX_ERR_ARG, "eigs_lowest: empty problem");
    const int max_it = opts->max_it > 0 ? opts->max_it : std::max<int>(100, (int)(2 * N / m));
    const double tol = opts->tol > 0 ? opts->tol : 1e-8;
    const auto t_begin = std::chrono::steady_clock::now();

    DevBuf dV, dW, dX, dTmp, dPartial, dScal, dQ;
    DMRGX_CHK(dV.alloc((size_t)(m + 1) * n * sizeof(double)));
    DMRGX_CHK(dW.alloc((size_t)n * sizeof(double)));
    DMRGX_CHK(dTmp.alloc((size_t)(m / 2 + 2) * n * sizeof(double)));
    if (dist) DMRGX_CHK(dX.alloc((size_t)I.vec_len * sizeof(double)));
    DMRGX_CHK(dPartial.alloc((size_t)(MAX_NCV + DOT_CHUNK + 1) * DOT_BLOCKS * sizeof(double)));
    // scalars: per step j a row of (m+2) doubles: h_j[0..m] accumulated coefficients, slot m+1: beta_j^2 ; + scratch c[]
    const int row = m + 2;
    DMRGX_CHK(dScal.alloc((size_t)((m + 1) * row + 3 * (MAX_NCV + 2)) * sizeof(double)));
    DMRGX_CHK(dQ.alloc((size_t)MAX_NCV * MAX_NCV * sizeof(double)));
    double* V = dV.as<double>();
    double* w = dW.as<double>();
    double* c1 = dScal.as<double>() + (size_t)(m + 1) * row;      // [nv+1]: V^T w, w.w
    double* c2 = c1 + (MAX_NCV + 2);                              // [nv+1]: V^T w', w'.w' (second pass)
    double* nrm = c2 + (MAX_NCV + 2);                             // [1]
    auto Hrow = [&](int j) { return dScal.as<double>() + (size_t)j * row; };
    auto vec = [&](int j) { return V + (size_t)j * n; };
    DMRGX_HIP(zero_async(dV.p, dV.bytes, st));        // (the Ritz-vector rotations read all m rows of V against zero-padded coefficients)
    DMRGX_HIP(zero_async(dW.p, dW.bytes, st));
    if (dist) DMRGX_HIP(zero_async(dX.p, dX.bytes, st));

    const bool vec2 = (n % 2 == 0);                     // all vectors are whole allocations (256-byte aligned) of stride n
    const int nblk = DOT_BLOCKS;      // (scaling the grid down with n was measured slower even at n = 1.6e5: these passes are latency-bound)
    auto allreduce = [&](double* buf, int64_t count) -> dmrgx_status {
        if (!dist) return DMRGX_OK;
        return hooks ? opts->allreduce_sum(opts->user, buf, count, st) : dmrgx_comm_allreduce_sum(opts->comm, buf, count, st);
    };
    // dots of w against V[0..nv) plus w.w  -> c1[0..nv]
    auto multi_dot = [&](int nv) -> dmrgx_status {
        const int chunks = (nv + 1 + DOT_CHUNK - 1) / DOT_CHUNK;
        if (vec2) hipLaunchKernelGGL(multi_dot_kernel<true>, dim3(nblk, chunks), dim3(DOT_THREADS), 0, st, V, n, nv, w, n, dPartial.as<double>());
        else hipLaunchKernelGGL(multi_dot_kernel<false>, dim3(nblk, chunks), dim3(DOT_THREADS), 0, st, V, n, nv, w, n, dPartial.as<double>());
        DMRGX_HIP(hipGetLastError());
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(nv + 1), dim3(DOT_THREADS), 0, st, dPartial.as<double>(), c1, nv + 1, nblk);
        DMRGX_HIP(hipGetLastError());
        return allreduce(c1, nv + 1);
    };
    // final == false: only subtract.  final == true: also set nrm = ||w_new||^2 from the dots just reduced.
    auto multi_axpy = [&](int nv, double* hacc, bool final) -> dmrgx_status {
        if (final) { hipLaunchKernelGGL(norm_after_projection_kernel, dim3(1), dim3(64), 0, st, c1, nv, nrm); DMRGX_HIP(hipGetLastError()); }
        hipLaunchKernelGGL(multi_axpy_kernel, dim3(nblk), dim3(DOT_THREADS), 0, st, V, n, nv, c1, w, n, dPartial.as<double>(), hacc);
        DMRGX_HIP(hipGetLastError());
        return DMRGX_OK;
    };
    auto normalise_into = [&](double* dst) -> dmrgx_status {
        hipLaunchKernelGGL(scale_copy_kernel, dim3(1024), dim3(256), 0, st, w, dst, n, nrm);
        DMRGX_HIP(hipGetLastError());
        return DMRGX_OK;
    };
    CommTimer ctimer(st);       // (distributed only) HIP events round the all-gather and the apply of every MatMult: dmrgx_eigs_comm_timing
    auto matvec = [&](const double* v_local, double* y_local) -> dmrgx_status {
        if (!dist) return dmrgx_kron_apply(plan, v_local, y_local, st);
        DMRGX_HIP(hipMemcpyAsync(dX.as<double>() + I.local_offset, v_local, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        ctimer.mark();
        DMRGX_CHK(gather_full(dX.as<double>()));
        ctimer.mark();
        const dmrgx_status rc = dmrgx_kron_apply(plan, dX.as<double>(), y_local, st);
        ctimer.mark();
        return rc;
    };

    // ---- start vector: w <- v0 ; V[0] = w/||w|| ------------------------------------------------------------
    if (opts->use_initial) DMRGX_HIP(hipMemcpyAsync(w, psi_full + I.local_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    else {
        // only real entries get random values: padding of a striped segment must stay zero, so fill through the
        // plan's layout conversion when striped
        if (!dist) { hipLaunchKernelGGL(random_fill_kernel, dim3(1024), dim3(256), 0, st, w, n, opts->seed, (int64_t)0); DMRGX_HIP(hipGetLastError()); }
        else {
            DevBuf ref;
            DMRGX_CHK(ref.alloc((size_t)N * sizeof(double)));
            hipLaunchKernelGGL(random_fill_kernel, dim3(1024), dim3(256), 0, st, ref.as<double>(), N, opts->seed, (int64_t)0);
            DMRGX_HIP(hipGetLastError());
            DMRGX_CHK(dmrgx_kron_vec_to_striped(plan, ref.as<double>(), dX.as<double>(), st));
            DMRGX_HIP(hipMemcpyAsync(w, dX.as<double>() + I.local_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
            DMRGX_HIP(hipStreamSynchronize(st));
        }
    }
    DMRGX_CHK(multi_dot(0));                                  // c1[0] = w.w
    double* const pin = opts->use_initial ? pinned_scalars() : nullptr;
    if (pin) { pin[2] = 0.0; DMRGX_HIP(hipMemcpyAsync(pin + 2, c1, sizeof(double), hipMemcpyDeviceToHost, st)); }      // read after the first synchronisation below
    DMRGX_HIP(hipMemcpyAsync(nrm, c1, sizeof(double), hipMemcpyDeviceToDevice, st));
    DMRGX_CHK(normalise_into(vec(0)));

    // how often the Ritz pair of a caller-supplied start vector is looked at: a look costs one small copy + a stream sync
    // (~50 us), a MatMult that turns out unnecessary 0.1 .. 10 ms
    // looks at the Ritz pair of a caller-supplied start vector: pinned slots + events of this host thread (see the step loop)
    // (also from a random start vector: SLEPc looks at restarts only, but a look costs nothing here and a solve then ends up to ncv / 2 - 1
    //  MatMults earlier -- the warm-up's cluster-growth steps and the reference-settings runs start random)
    const bool looks = opts->max_matvec <= 0;
    static thread_local double* look_buf = nullptr;
    static thread_local hipEvent_t look_ev[2] = {nullptr, nullptr};
    constexpr size_t look_stride = (size_t)(MAX_NCV + 1) * (MAX_NCV + 2);
    if (looks && !look_buf) {
        DMRGX_HIP(hipHostMalloc((void**)&look_buf, 2 * look_stride * sizeof(double), hipHostMallocDefault));
        DMRGX_HIP(hipEventCreateWithFlags(&look_ev[0], hipEventDisableTiming)); DMRGX_HIP(hipEventCreateWithFlags(&look_ev[1], hipEventDisableTiming));
    }
    int look_mm[2] = {0, 0};                    // columns covered by the look in flight in each slot (0: none)
    std::vector<double> T((size_t)m * m, 0.0), theta, Q, hbuf((size_t)(m + 1) * row);
    int k = 0, n_matvec = 0, restarts = 0, converged = 0;
    double beta_m = 0.0, resid = 0.0, lambda = 0.0;
    std::vector<double> Qdev;
    // A caller-supplied start vector of zero (or NaN) norm is normalised to the zero vector (scale 0 instead of 1 / 0), every Krylov
    // vector after it is zero and the projected matrix is exactly zero: "converged" at E = 0 with psi = 0.  Seen from the first
    // coefficients that come back to the host -- alpha_0 and beta_0^2 both not positive in magnitude -- the solve is repeated from the
    // random start vector instead (the engine's projected start vectors can lose all their weight).
    auto null_start = [&]() { return opts->use_initial && restarts == 0 && k == 0 && ((!(std::fabs(hbuf[0]) > 0.0) && !(hbuf[(size_t)m + 1] > 0.0)) || start_too_light(opts, pin)); };
    auto redo_from_random = [&]() -> dmrgx_status {
        // (called right after a synchronisation: nothing is queued on the work space, which the repeated solve allocates again -- ADVICE
        //  round 4: the rejected attempt's buffers are released first and its MatMults and seconds stay in the returned totals)
        dV.release(); dW.release(); dTmp.release(); dX.release();
        dmrgx_eigs_opts o2 = *opts;
        o2.use_initial = 0;
        dmrgx_eigs_stats s2;
        memset(&s2, 0, sizeof(s2));
        const dmrgx_status rc = dmrgx_eigs_lowest(plan, &o2, e0, psi_full, &s2, (void*)st);
        if (stats) { *stats = s2; stats->n_matvec += n_matvec; stats->start_rejected = 1;
                     stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); }
        return rc;
    };
    while (true) {
        DMRGX_HIP(zero_async(Hrow(k), (size_t)(m + 1 - k) * row * sizeof(double), st));
        int jend = m;                                         // benchmark mode: stop after exactly max_matvec MatMults
        if (opts->max_matvec > 0) jend = std::min(m, k + std::max(0, opts->max_matvec - n_matvec));
        const bool capped = jend < m || (opts->max_matvec > 0 && n_matvec + (m - k) >= opts->max_matvec);
        bool early = false;
        for (int j = k; j < jend; ++j) {
            DMRGX_CHK(matvec(vec(j), w));
            ++n_matvec;
            const int nv = j + 1;
            DMRGX_CHK(multi_dot(nv));                         // pass 1: c1 = V^T w                (one fused all-reduce)
            if (nv <= FUSE_NV) {
                // w' = w - V c1 fused with pass 2's dots: c2 = V^T w', w'.w'            (one fused all-reduce)
                hipLaunchKernelGGL(axpy_dot_kernel<FUSE_NV>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, n, nv, c1, w, n, dPartial.as<double>(), Hrow(j));
                DMRGX_HIP(hipGetLastError());
                hipLaunchKernelGGL(reduce_partials_kernel, dim3(nv + 1), dim3(DOT_THREADS), 0, st, dPartial.as<double>(), c2, nv + 1, nblk);
                DMRGX_HIP(hipGetLastError());
                DMRGX_CHK(allreduce(c2, nv + 1));
            } else {
                DMRGX_CHK(multi_axpy(nv, Hrow(j), false));
                DMRGX_CHK(multi_dot(nv));
                DMRGX_HIP(hipMemcpyAsync(c2, c1, (size_t)(nv + 1) * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            // beta^2 = w'.w' - |c2|^2 ; v_{j+1} = (w' - V c2) / beta
            if (vec2) hipLaunchKernelGGL(axpy_normalise_kernel<true>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, n, nv, c2, w, vec(j + 1), n, Hrow(j) + m + 1, Hrow(j));
            else hipLaunchKernelGGL(axpy_normalise_kernel<false>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, n, nv, c2, w, vec(j + 1), n, Hrow(j) + m + 1, Hrow(j));
            DMRGX_HIP(hipGetLastError());
            // A start vector supplied by the caller (the sweep engine's transformed ground state) is usually within a few Lanczos steps of
            // convergence, so the Ritz pair is looked at after EVERY step -- one step behind the queue (round 5; rounds 2-4 looked every
            // 1 / 2 / 4 steps with a blocking copy: 0.24 ms of idle GPU per configs[1] step and, on the small superblocks, 1-2 MatMults past
            // convergence): the coefficients of step j travel to pinned memory behind an event, and the host reads the look of step j - 1
            // while step j runs.  Convergence seen there ends the solve with the step already queued included.
            if (looks && j + 1 < m) {
                const int slot = j & 1;
                DMRGX_HIP(hipMemcpyAsync(look_buf + (size_t)slot * look_stride, dScal.p, (size_t)(m + 1) * row * sizeof(double), hipMemcpyDeviceToHost, st));
                DMRGX_HIP(hipEventRecord(look_ev[slot], st));
                look_mm[slot] = j + 1;
            }
            if (looks && j > k && look_mm[(j - 1) & 1] == j) {
                const int slot = (j - 1) & 1;
                look_mm[slot] = 0;
                DMRGX_HIP(hipEventSynchronize(look_ev[slot]));
                std::copy(look_buf + (size_t)slot * look_stride, look_buf + (size_t)slot * look_stride + (size_t)(m + 1) * row, hbuf.begin());
                if (null_start()) { DMRGX_HIP(hipStreamSynchronize(st)); return redo_from_random(); }
                const int mm = j;
                std::vector<double> A((size_t)mm * mm, 0.0), th, Qs;
                for (int jj = 0; jj < mm; ++jj) for (int i = 0; i <= jj; ++i) {
                    const double v = (jj >= k) ? hbuf[(size_t)jj * row + i] : T[(size_t)i * m + jj];
                    A[(size_t)i * mm + jj] = v; A[(size_t)jj * mm + i] = v;
                }
                jacobi_eigh(mm, A, th, Qs);
                const double r = std::fabs(std::sqrt(std::max(0.0, hbuf[(size_t)(mm - 1) * row + m + 1])) * Qs[(size_t)(mm - 1) * mm + 0]);
                static const bool trace = getenv("DMRGX_EIGS_TRACE") != nullptr;      // developer aid: Ritz value and residual estimate per look
                if (trace) fprintf(stderr, "[eigs] look at matvec %d (queued: %d): theta %.12f  |r| %.3e  (target %.3e)\n", n_matvec - 1, n_matvec, th[0], r, tol * std::fabs(th[0]));
                if (r <= tol * std::max(std::fabs(th[0]), 1e-300)) { jend = j + 1; early = true; break; }      // (step j is queued: its column is part of the final Rayleigh-Ritz)
            }
        }
        look_mm[0] = look_mm[1] = 0;
        DMRGX_HIP(hipMemcpyAsync(hbuf.data(), dScal.p, (size_t)(m + 1) * row * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        if (null_start()) return redo_from_random();
        if (jend < m) {                                       // truncated cycle: Rayleigh-Ritz on the jend columns built so far
            const int mm = jend;
            if (mm < 1) break;
            std::vector<double> A((size_t)mm * mm, 0.0), th, Qs;
            for (int j = 0; j < mm; ++j) for (int i = 0; i <= j; ++i) {
                const double v = (j >= k) ? hbuf[(size_t)j * row + i] : T[(size_t)i * m + j];
                A[(size_t)i * mm + j] = v; A[(size_t)j * mm + i] = v;
            }
            jacobi_eigh(mm, A, th, Qs);
            lambda = th[0];
            resid = std::fabs(std::sqrt(std::max(0.0, hbuf[(size_t)(mm - 1) * row + m + 1])) * Qs[(size_t)(mm - 1) * mm + 0]);
            Q.assign((size_t)m * m, 0.0);
            for (int i = 0; i < mm; ++i) Q[(size_t)i * m + 0] = Qs[(size_t)i * mm + 0];
            ++restarts;
            if (early) converged = 1;
            break;
        }
        // projected matrix: kept Ritz block is diagonal, new columns come from the recorded coefficients
        for (int j = k; j < m; ++j)
            for (int i = 0; i <= j; ++i) { T[(size_t)i * m + j] = hbuf[(size_t)j * row + i]; T[(size_t)j * m + i] = T[(size_t)i * m + j]; }
        beta_m = std::sqrt(std::max(0.0, hbuf[(size_t)(m - 1) * row + m + 1]));
        std::vector<double> A = T;
        jacobi_eigh(m, A, theta, Q);
        lambda = theta[0];
        resid = std::fabs(beta_m * Q[(size_t)(m - 1) * m + 0]);
        ++restarts;
        if (resid <= tol * std::max(std::fabs(lambda), 1e-300) || m == N) { converged = 1; break; }
        if (restarts >= max_it || capped) break;
        // thick restart: keep the kk lowest Ritz vectors + the residual direction V[m]
        const int kk = std::max(1, std::min(m / 2, m - 1));
        Qdev.assign((size_t)m * kk, 0.0);
        for (int i = 0; i < m; ++i) for (int j = 0; j < kk; ++j) Qdev[(size_t)i * kk + j] = Q[(size_t)i * m + j];
        DMRGX_HIP(h2d_async(dQ.p, Qdev.data(), Qdev.size() * sizeof(double), st));
        hipLaunchKernelGGL(basis_rotate_kernel, dim3(1024, (kk + DOT_CHUNK - 1) / DOT_CHUNK), dim3(DOT_THREADS), 0, st,
                           V, n, m, dQ.as<double>(), kk, kk, dTmp.as<double>(), n, n);
        DMRGX_HIP(hipGetLastError());
        DMRGX_HIP(hipMemcpyAsync(V, dTmp.p, (size_t)kk * n * sizeof(double), hipMemcpyDeviceToDevice, st));
        DMRGX_HIP(hipMemcpyAsync(vec(kk), vec(m), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        DMRGX_HIP(hipStreamSynchronize(st));                  // Qdev is reused by the next cycle
        std::fill(T.begin(), T.end(), 0.0);
        for (int i = 0; i < kk; ++i) T[(size_t)i * m + i] = theta[i];
        k = kk;
    }
    // ---- eigenvector: psi = V_m Q[:,0], renormalised -------------------------------------------------------
    Qdev.assign((size_t)m, 0.0);
    for (int i = 0; i < m; ++i) Qdev[i] = Q[(size_t)i * m + 0];
    DMRGX_HIP(h2d_async(dQ.p, Qdev.data(), Qdev.size() * sizeof(double), st));
    hipLaunchKernelGGL(basis_rotate_kernel, dim3(1024, 1), dim3(DOT_THREADS), 0, st, V, n, m, dQ.as<double>(), 1, 1, w, n, n);
    DMRGX_HIP(hipGetLastError());
    DMRGX_CHK(multi_dot(0));
    DMRGX_HIP(hipMemcpyAsync(nrm, c1, sizeof(double), hipMemcpyDeviceToDevice, st));
    if (!dist) DMRGX_CHK(normalise_into(psi_full));
    else {
        DMRGX_CHK(normalise_into(psi_full + I.local_offset));
        DMRGX_CHK(gather_full(psi_full));
    }
    DMRGX_HIP(hipStreamSynchronize(st));
    *e0 = lambda;
    if (stats) {
        stats->n_matvec = n_matvec; stats->n_restart = restarts; stats->converged = converged; stats->start_rejected = 0; stats->residual = resid;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    if (!converged) DMRGX_FAIL(DMRGX_ERR_NOTCONV, "eigs_lowest: not converged after %d restarts (residual %.3e)", restarts, resid);
    return DMRGX_OK;
}


// ---- generalized Davidson (opts->method == 1) ---------------------------------------------------------------------------------
// Basis V (orthonormal), W = H V, projected matrix G = V^T W.  Per iteration: one MatMult (the new basis vector), the new column of G
// (one fused dot pass), the lowest Ritz pair of G (gd_ritz_kernel, on the device), then ONE pass that forms the Ritz vector, the true
// residual r = H u - theta u, the diagonally preconditioned correction t = r / (theta - diag H) (dmrgx_kron_diag) and the first
// Gram-Schmidt dots V^T t, which is then orthogonalised against V by the CGS2 kernels of the Lanczos path.
// Round 5: a SMALL search space with a GD+k restart.  The vector work of an iteration is 4 j + 7 passes over vectors for a basis of j
// (5 j + 7 before the dots were fused), so the basis is held at gd_basis <= 8 vectors and restarted to the lowest Ritz vector(s) plus
// the previous iteration's Ritz vector -- the restart that keeps the convergence of the unrestarted method (Stathopoulos & Saad;
// measured on CPU-side model problems, tests/experiments/gd_restart_basis.py: 12.9 MatMults per solve for every maximal basis from 24 down to 4).
// Distributed like the Lanczos path: vectors are stripe segments, every dot is all-reduced, x is all-gathered before a MatMult.
static dmrgx_status eigs_davidson(dmrgx_kron_plan* plan, const dmrgx_eigs_opts* opts, double* e0, double* psi_full, dmrgx_eigs_stats* stats, hipStream_t st)
{
    dmrgx_kron_info I;
    DMRGX_CHK(dmrgx_kron_plan_info(plan, &I));
    const bool dist = I.vec_len != I.n_states;
    const bool hooks = opts->allgather && opts->allreduce_sum;
    if (dist && !hooks && !opts->comm) DMRGX_FAIL(DMRGX_ERR_ARG, "eigs_lowest: a striped plan needs a communicator (opts->comm) or the allgather/allreduce hooks");
    const int64_t n = I.local_len, N = I.n_states;
    // Every work vector is padded to an EVEN length ld with a zero in the pad: basis rows are then 16-byte aligned whatever n is, and the
    // vector kernels run their two-doubles-per-lane form on odd n as well (half of all steps; multi_dot 42 -> 30 us at m = 2048).  The pad
    // stays zero by construction: every kernel writes it as a combination of zeros, the MatMult never touches it.
    const int64_t ld = (n + 1) & ~(int64_t)1;
    // search space: opts->ncv vectors at most (default GD_NV = 8: SLEPc's meaning of ncv for its gd solver), restarted to gd_minv Ritz
    // vectors + the previous Ritz vector.  A basis that could not grow after a restart (fewer than minv + 2 vectors) is widened.
    int m = opts->ncv > 0 ? opts->ncv : GD_NV;
    m = (int)std::min<int64_t>(std::min(m, GD_MAX - 1), N);
    if (m < 2) DMRGX_FAIL(DMRGX_ERR_ARG, "eigs_lowest (gd): needs at least a two-dimensional search space");
    if (m < 3 && N >= 3) m = 3;
    const int keep = std::max(1, std::min(opts->gd_minv > 0 ? opts->gd_minv : 1, m - 2));
    const int kk = keep + 1;                   // basis after a restart
    const int max_mv = opts->max_it > 0 ? opts->max_it * 16 : std::max<int>(1600, (int)(2 * N));
    const int hand_over = 96;                  // MatMults after which a start vector that was not close after all goes to the Lanczos path
    const double tol = opts->tol > 0 ? opts->tol : 1e-8;
    const auto t_begin = std::chrono::steady_clock::now();
    const int nblk = DOT_BLOCKS;
    const bool fused = m <= GD_NV;             // the correction kernel also takes the first Gram-Schmidt dots

    static const bool trace_setup = getenv("DMRGX_EIGS_TRACE") != nullptr;      // developer aid: wall time of the set-up stages (synchronising)
    auto mark = [&](const char* what) {
        if (!trace_setup) return;
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[eigs gd] setup %-12s t = %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    mark("enter");
    DevBuf dV, dW, dT, dX, dD, dTmp, dPartial, dScal, dState;
    DMRGX_CHK(dV.alloc((size_t)(m + 1) * ld * sizeof(double)));
    DMRGX_CHK(dW.alloc((size_t)(m + 1) * ld * sizeof(double)));
    DMRGX_CHK(dT.alloc((size_t)ld * sizeof(double)));
    DMRGX_CHK(dD.alloc((size_t)ld * sizeof(double)));
    DMRGX_CHK(dTmp.alloc((size_t)kk * ld * sizeof(double)));
    if (dist) DMRGX_CHK(dX.alloc((size_t)I.vec_len * sizeof(double)));
    DMRGX_CHK(dPartial.alloc((size_t)(MAX_NCV + DOT_CHUNK + 2) * DOT_BLOCKS * sizeof(double)));
    DMRGX_CHK(dScal.alloc((size_t)(3 * (MAX_NCV + 2)) * sizeof(double)));
    DMRGX_CHK(dState.alloc(sizeof(GdState)));
    double* V = dV.as<double>();
    double* W = dW.as<double>();
    double* t = dT.as<double>();
    double* c1 = dScal.as<double>();
    double* c2 = c1 + (MAX_NCV + 2);
    double* nrm = c2 + (MAX_NCV + 2);
    GdState* S = dState.as<GdState>();
    const double* const S_y = (const double*)((const char*)dState.p + offsetof(GdState, y));
    const double* const S_e0 = (const double*)((const char*)dState.p + offsetof(GdState, e0));
    const double* const S_Q = (const double*)((const char*)dState.p + offsetof(GdState, Q));
    auto vec = [&](int j) { return V + (size_t)j * ld; };
    auto wvec = [&](int j) { return W + (size_t)j * ld; };
    // (every basis vector is written whole before it is read; only the padding of a striped segment relies on the initial zeros)
    if (dist) { DMRGX_HIP(zero_async(dV.p, dV.bytes, st)); DMRGX_HIP(zero_async(dW.p, dW.bytes, st)); }
    if (dist) DMRGX_HIP(zero_async(dT.p, dT.bytes, st));
    DMRGX_HIP(zero_async(dState.p, dState.bytes, st));
    if (dist) DMRGX_HIP(zero_async(dX.p, dX.bytes, st));
    mark("buffers");
    DMRGX_CHK(dmrgx_kron_diag(plan, dD.as<double>(), st));
    if (ld != n) {              // the pads nobody writes: the MatMult's outputs (W), the start vector's copy (t), the diagonal
        hipLaunchKernelGGL(gd_zero_pads_kernel, dim3(1), dim3(64), 0, st, W, ld, m + 1, n, t, dD.as<double>());
        DMRGX_HIP(hipGetLastError());
    }
    mark("diagonal");
    auto allreduce = [&](double* buf, int64_t count) -> dmrgx_status {
        if (!dist) return DMRGX_OK;
        return hooks ? opts->allreduce_sum(opts->user, buf, count, st) : dmrgx_comm_allreduce_sum(opts->comm, buf, count, st);
    };
    auto gather_full = [&](double* full) -> dmrgx_status {
        return hooks ? opts->allgather(opts->user, full, I.seg_stride, st) : dmrgx_comm_allgather(opts->comm, full, I.seg_stride, st);
    };
    // per-block partial sums of V[0..nv)^T x and x.x (dPartial); reduce == true: also out[0..nv] = the sums (all-reduced)
    auto multi_dot = [&](int nv, const double* x, double* out, bool reduce) -> dmrgx_status {
        const int chunks = (nv + 1 + DOT_CHUNK - 1) / DOT_CHUNK;
        hipLaunchKernelGGL(multi_dot_kernel<true>, dim3(nblk, chunks), dim3(DOT_THREADS), 0, st, V, ld, nv, x, ld, dPartial.as<double>());
        DMRGX_HIP(hipGetLastError());
        if (!reduce) return DMRGX_OK;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(nv + 1), dim3(DOT_THREADS), 0, st, dPartial.as<double>(), out, nv + 1, nblk);
        DMRGX_HIP(hipGetLastError());
        return allreduce(out, nv + 1);
    };
    // second half of CGS2 on x, whose first-pass dots c1 = V^T x are reduced: dst = (x - V V^T x) / || . ||; x is overwritten
    auto orthonormalise_tail = [&](int nv, double* x, double* dst) -> dmrgx_status {
        if (nv <= GD_NV) hipLaunchKernelGGL(axpy_dot_kernel<GD_NV>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, ld, nv, c1, x, ld, dPartial.as<double>(), (double*)nullptr);
        else hipLaunchKernelGGL(axpy_dot_kernel<FUSE_NV>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, ld, nv, c1, x, ld, dPartial.as<double>(), (double*)nullptr);
        DMRGX_HIP(hipGetLastError());
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(nv + 1), dim3(DOT_THREADS), 0, st, dPartial.as<double>(), c2, nv + 1, nblk);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(allreduce(c2, nv + 1));
        hipLaunchKernelGGL(axpy_normalise_kernel<true>, dim3(nblk), dim3(DOT_THREADS), 0, st, V, ld, nv, c2, x, dst, ld, (double*)nullptr, (double*)nullptr);
        DMRGX_HIP(hipGetLastError());
        return DMRGX_OK;
    };
    CommTimer ctimer(st);       // (distributed only) HIP events round the all-gather and the apply of every MatMult: dmrgx_eigs_comm_timing
    auto matvec = [&](const double* v_local, double* y_local) -> dmrgx_status {
        if (!dist) return dmrgx_kron_apply(plan, v_local, y_local, st);
        DMRGX_HIP(hipMemcpyAsync(dX.as<double>() + I.local_offset, v_local, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        ctimer.mark();
        DMRGX_CHK(gather_full(dX.as<double>()));
        ctimer.mark();
        const dmrgx_status rc = dmrgx_kron_apply(plan, dX.as<double>(), y_local, st);
        ctimer.mark();
        return rc;
    };

    // ---- start vector ----------------------------------------------------------------------------------------------------
    if (opts->use_initial) DMRGX_HIP(hipMemcpyAsync(t, psi_full + I.local_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    else if (!dist) { hipLaunchKernelGGL(random_fill_kernel, dim3(1024), dim3(256), 0, st, t, n, opts->seed, (int64_t)0); DMRGX_HIP(hipGetLastError()); }
    else {
        DevBuf ref;
        DMRGX_CHK(ref.alloc((size_t)N * sizeof(double)));
        hipLaunchKernelGGL(random_fill_kernel, dim3(1024), dim3(256), 0, st, ref.as<double>(), N, opts->seed, (int64_t)0);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(dmrgx_kron_vec_to_striped(plan, ref.as<double>(), dX.as<double>(), st));
        DMRGX_HIP(hipMemcpyAsync(t, dX.as<double>() + I.local_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        DMRGX_HIP(hipStreamSynchronize(st));
    }
    DMRGX_CHK(multi_dot(0, t, c1, true));
    DMRGX_CHK(orthonormalise_tail(0, t, vec(0)));
    double* const pin = opts->use_initial ? pinned_scalars() : nullptr;
    if (pin) { pin[2] = 0.0; DMRGX_HIP(hipMemcpyAsync(pin + 2, c2, sizeof(double), hipMemcpyDeviceToHost, st)); }      // c2[0] = |start vector|^2; read at the first look
    mark("start vector");

    // the looks come back through pinned memory behind an event, so that the orthogonalisation of the next direction is already queued
    // when the host waits for them; one buffer + event per host thread (callers on different threads / devices do not share them)
    static thread_local double* h_look = nullptr;
    static thread_local hipEvent_t ev_look = nullptr;
    if (!h_look) { DMRGX_HIP(hipHostMalloc((void**)&h_look, 64, hipHostMallocDefault)); DMRGX_HIP(hipEventCreateWithFlags(&ev_look, hipEventDisableTiming)); }
    DevBuf dLook;
    DMRGX_CHK(dLook.alloc(4 * sizeof(double)));
    int j = 0, n_matvec = 0, restarts = 0, converged = 0;
    double lambda = 0.0, resid = 0.0;
    const double floor_rel = 1e-3;
    bool warm = false;                          // the device state holds the Ritz basis of the previous iteration in the current basis
    bool rotated = false;                       // a restart has been queued since the last Ritz solve: the Ritz vector is the first basis vector
    int mm = 0;
    while (true) {
        DMRGX_CHK(matvec(vec(j), wvec(j)));
        ++n_matvec;
        mm = j + 1;
        const bool restart = mm == m;
        DMRGX_CHK(multi_dot(mm, wvec(j), c1, dist));                               // column j of G = V^T w_j (reduced by the Ritz kernel itself unless an all-reduce must come first)
        hipLaunchKernelGGL(gd_ritz_kernel, dim3(1), dim3(RITZ_THREADS), 0, st, S, dist ? (const double*)c1 : (const double*)dPartial.as<double>(), dist ? 0 : 1, nblk, mm, warm ? 1 : 0, restart ? 1 : 0, keep);
        DMRGX_HIP(hipGetLastError());
        warm = !restart;
        rotated = false;
        // correction vector (+ first Gram-Schmidt dots) and |r|^2
        const int look_idx = fused ? mm + 1 : 0;
        double* red_out = fused ? c1 : nrm;
        if (fused) hipLaunchKernelGGL((ritz_precond_kernel<GD_NV, true>), dim3(nblk), dim3(DOT_THREADS), 0, st, (const double*)V, (const double*)W, ld, mm, (const GdState*)S,
                                      (const double*)dD.as<double>(), floor_rel, t, ld, dPartial.as<double>());
        else hipLaunchKernelGGL((ritz_precond_kernel<FUSE_NV, false>), dim3(nblk), dim3(DOT_THREADS), 0, st, (const double*)V, (const double*)W, ld, mm, (const GdState*)S,
                                (const double*)dD.as<double>(), floor_rel, t, ld, dPartial.as<double>());
        DMRGX_HIP(hipGetLastError());
        const int nred = fused ? mm + 2 : 1;
        hipLaunchKernelGGL(gd_reduce_look_kernel, dim3(nred), dim3(DOT_THREADS), 0, st, (const double*)dPartial.as<double>(), red_out, nred, nblk, look_idx, 0, (const GdState*)S, dLook.as<double>());
        DMRGX_HIP(hipGetLastError());
        if (dist) {
            DMRGX_CHK(allreduce(red_out, nred));
            hipLaunchKernelGGL(gd_reduce_look_kernel, dim3(nred), dim3(DOT_THREADS), 0, st, (const double*)dPartial.as<double>(), red_out, nred, nblk, look_idx, 1, (const GdState*)S, dLook.as<double>());
            DMRGX_HIP(hipGetLastError());
        }
        DMRGX_HIP(hipMemcpyAsync(h_look, dLook.p, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipEventRecord(ev_look, st));
        // queued behind the look, before the host reads it: the restart (when the basis is full) and the next direction
        const bool may_continue = mm < N && n_matvec < max_mv && n_matvec < hand_over;
        if (may_continue) {
            bool have_dots = fused;
            if (restart) {
                // GD+k restart: V <- V Q, W <- W Q with Q = [lowest Ritz vectors | previous Ritz vector] from the device state; G <- Q^T G Q there
                for (double* B : {V, W}) {
                    hipLaunchKernelGGL(basis_rotate_kernel, dim3(1024, (kk + DOT_CHUNK - 1) / DOT_CHUNK), dim3(DOT_THREADS), 0, st, (const double*)B, ld, m, S_Q, kk, kk,
                                       dTmp.as<double>(), ld, ld);
                    DMRGX_HIP(hipGetLastError());
                    DMRGX_HIP(hipMemcpyAsync(B, dTmp.p, (size_t)kk * ld * sizeof(double), hipMemcpyDeviceToDevice, st));
                }
                j = kk - 1;
                ++restarts;
                rotated = true;
                have_dots = false;              // (the dots were taken against the old basis)
            }
            if (!have_dots) DMRGX_CHK(multi_dot(j + 1, t, c1, true));
            DMRGX_CHK(orthonormalise_tail(j + 1, t, vec(j + 1)));                 // next basis vector from the preconditioned residual
        }
        DMRGX_HIP(hipEventSynchronize(ev_look));
        resid = std::sqrt(std::max(h_look[0], 0.0));
        lambda = h_look[1];
        if (n_matvec == 1 && (!(h_look[2] > 0.0) || start_too_light(opts, pin))) {      // |H v_0|^2 is not positive: the start vector had zero (or NaN) norm -- see the Lanczos path; or it is lighter than the caller accepts
            DMRGX_HIP(hipStreamSynchronize(st));
            dV.release(); dW.release(); dTmp.release(); dT.release(); dD.release();      // (the repeated solve allocates its own work space)
            dmrgx_eigs_opts o2 = *opts;
            o2.method = 0; o2.use_initial = 0;
            dmrgx_eigs_stats s2;
            memset(&s2, 0, sizeof(s2));
            const dmrgx_status rc = dmrgx_eigs_lowest(plan, &o2, e0, psi_full, &s2, (void*)st);
            if (stats) { *stats = s2; stats->n_matvec += n_matvec; stats->start_rejected = 1;
                         stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); }
            return rc;
        }
        static const bool trace = getenv("DMRGX_EIGS_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[eigs gd] matvec %d: basis %d theta %.12f  |r| %.3e  (target %.3e)  t = %.3f ms\n", n_matvec, mm, lambda, resid, tol * std::fabs(lambda),
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
        if (resid <= tol * std::max(std::fabs(lambda), 1e-300) || mm == N) { converged = 1; break; }
        if (n_matvec >= max_mv) break;
        if (n_matvec >= hand_over) {
            // a start vector that was not close after all (the preconditioned iteration then trails Lanczos, see above): hand the
            // current Ritz vector to the Lanczos path instead of iterating on
            hipLaunchKernelGGL(basis_rotate_kernel, dim3(1024, 1), dim3(DOT_THREADS), 0, st, (const double*)V, ld, mm, S_y, 1, 1,
                               dist ? psi_full + I.local_offset : psi_full, n, n);
            DMRGX_HIP(hipGetLastError());
            DMRGX_HIP(hipStreamSynchronize(st));
            dV.release(); dW.release(); dTmp.release(); dT.release(); dD.release();
            dmrgx_eigs_opts o2 = *opts;
            o2.method = 0; o2.use_initial = 1; o2.min_initial_norm2 = 0.0;
            dmrgx_eigs_stats s2;
            memset(&s2, 0, sizeof(s2));
            const dmrgx_status rc = dmrgx_eigs_lowest(plan, &o2, e0, psi_full, &s2, (void*)st);
            if (stats) { *stats = s2; stats->n_matvec += n_matvec; stats->n_restart += restarts;
                         stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); }
            return rc;
        }
        ++j;
    }
    // ---- eigenvector: psi = V y, renormalised --------------------------------------------------------------------------------
    {
        // the Ritz vector: V y in the basis the Ritz problem was solved in -- or, when a restart has been queued since, simply the
        // first vector of the new basis
        const int nb = rotated ? 1 : mm;
        hipLaunchKernelGGL(basis_rotate_kernel, dim3(1024, 1), dim3(DOT_THREADS), 0, st, (const double*)V, ld, nb, rotated ? S_e0 : S_y, 1, 1, t, ld, ld);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(multi_dot(0, t, c1, true));
        DMRGX_HIP(hipMemcpyAsync(nrm, c1, sizeof(double), hipMemcpyDeviceToDevice, st));
        double* dst = dist ? psi_full + I.local_offset : psi_full;
        hipLaunchKernelGGL(scale_copy_kernel, dim3(1024), dim3(256), 0, st, (const double*)t, dst, n, (const double*)nrm);
        DMRGX_HIP(hipGetLastError());
        if (dist) DMRGX_CHK(gather_full(psi_full));
        DMRGX_HIP(hipStreamSynchronize(st));
    }
    *e0 = lambda;
    if (stats) {
        stats->n_matvec = n_matvec; stats->n_restart = restarts; stats->converged = converged; stats->start_rejected = 0; stats->residual = resid;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    if (getenv("DMRGX_EIGS_TRACE")) fprintf(stderr, "[eigs gd] done: %d MatMults, %.3f ms\n", n_matvec, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    if (!converged) DMRGX_FAIL(DMRGX_ERR_NOTCONV, "eigs_lowest (gd): not converged after %d MatMults (residual %.3e)", n_matvec, resid);
    return DMRGX_OK;
}
