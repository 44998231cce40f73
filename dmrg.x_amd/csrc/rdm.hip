// K3/K4: reduced density matrices of the superblock ground state and their full eigendecomposition.
//
// Replaces the rank-0 loop of GetTruncation (reference include/DMRGBlockContainer.hpp:1715-1775): for every
// KronBlock k, Psi = n_L x n_R row-major slice of psi (:1731), rho_L = Psi Psi^T, rho_R = Psi^T Psi (:1733-1734),
// then EigRDM_BlockDiag = all eigenpairs of each block by LAPACK (:1962-2003).  Here all 2*nblocks matrices are
// built by ONE launch of the grouped MFMA-f64 GEMM and diagonalised TOGETHER on the device by a batched two-sided
// block-Jacobi method (block size JB = 16): per round every matrix contributes nb/2 disjoint block pairs; a 32 x 32
// sub-problem is solved by scalar Jacobi in LDS, then the rotation is applied to the block columns of A and V
// and to the block rows of A.  Sorting, the global m-cut and the stable re-sort by sector (:1795,1850-1853) stay
// on the host (caller), which fetches spectra with dmrgx_rdm_eigenvalues and asks for the kept eigenvectors with
// dmrgx_rdm_eigenvectors (== FillRotation_BlockDiag, :2006-2057).
#include "ggemm.h"
#include "hqr.h"
#include "symeig.h"
#include <chrono>
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

namespace dmrgx {
namespace {

#ifndef DMRGX_JB
#define DMRGX_JB 16
#endif
constexpr int JB = DMRGX_JB, JS = 2 * JB;          // block size, sub-problem size.  Every outer round costs ONE launch of pure
                                             // latency (~25 us at any matrix size: the sub-solves of the next round next to the
                                             // updates of this one); 32 x 32 sub-problems (31 dependent rotation rounds on 256
                                             // threads) measured best: JB = 32 halves the rounds but its 63-round, 1024-thread
                                             // solve is > 2x slower (one cyclic sweep per visit: the outer sweeps finish the job)
static_assert(JB == 16 || JB == 32, "sub-problem of 32 or 64 rows: (JS/16)^2 waves per workgroup");
constexpr int JLD = JS + 1;
constexpr int SUB_THREADS = JB * JB;           // sub-solve: one thread per pair of rotation pairs

struct MatDesc { int64_t a_off, a2_off, v_off; int32_t n, npad, nb, pad; };    // A lives in two buffers (a_off, a2_off): a round reads one, writes the other
struct PairRef { int32_t mat, j; };

__device__ __forceinline__ void pair_blocks(int nb, int r, int j, int& I, int& J)
{
    const int m1 = nb - 1;
    if (m1 == 0) { I = 0; J = 0; return; }
    const int rr = r % m1;
    if (j == 0) { I = m1; J = rr; }
    else { I = (rr + j) % m1; J = (rr - j + m1) % m1; }
    if (I > J) { const int t = I; I = J; J = t; }
}

// inverse of pair_blocks: the pair j of round r that block b sits in, and whether it is that pair's first (0) or second (1) block
__device__ __forceinline__ void block_seat(int nb, int r, int b, int& j, int& half)
{
    const int m1 = nb - 1;
    if (m1 == 0) { j = 0; half = 0; return; }
    const int rr = r % m1;
    if (b == m1 || b == rr) j = 0;
    else { const int d = (b - rr + m1) % m1; j = d <= (m1 - 1) / 2 ? d : m1 - d; }
    int I, J;
    pair_blocks(nb, r, j, I, J);
    half = (b == I) ? 0 : 1;
}

// A <- 0 on the padding (it stays exactly decoupled: rotations with a_pq == 0 are skipped, so the real eigenvectors are
// the columns [0,n) of V and the convergence norms only see the real matrix); V <- identity
__global__ void rdm_init_kernel(const MatDesc* __restrict__ mats, double* __restrict__ buf)
{
    const MatDesc m = mats[blockIdx.y];
    const int64_t tot = (int64_t)m.npad * m.npad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / m.npad), j = (int)(e % m.npad);
        buf[m.v_off + e] = (i == j) ? 1.0 : 0.0;
        if (i >= m.n || j >= m.n) buf[m.a_off + e] = 0.0;       // (the second buffer is written whole by the first round)
    }
}

struct TrTile { int64_t src, dst; int32_t nr, nc, ti, tj; };   // dst (nc x nr) = src (nr x nc)^T
__global__ void __launch_bounds__(256) transpose_kernel(const TrTile* __restrict__ tiles, const double* __restrict__ in, double* __restrict__ out)
{
    __shared__ double t[32][33];
    const TrTile k = tiles[blockIdx.x];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = k.ti * 32 + r, j = k.tj * 32 + tx;
        t[r][tx] = (i < k.nr && j < k.nc) ? in[k.src + (int64_t)i * k.nc + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = k.tj * 32 + r, i = k.ti * 32 + tx;
        if (i < k.nr && j < k.nc) out[k.dst + (int64_t)j * k.nr + i] = t[tx][r];
    }
}

// 1/sqrt(x): hardware estimate (v_rsq_f64) + two Newton steps -- the sub-solve is a chain of dependent scalar math on
// one wave per SIMD, so the length of this sequence is what a Jacobi round costs.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Jacobi rotation annihilating a_pq: with d = (a_qq - a_pp)/2 and h = hypot(d, a_pq), tan(theta) = sign(d) a_pq / (|d| + h)
// (the smaller root), c = 1/sqrt(1 + t^2), s = t c.  Division-free and as short as the dependency chain gets (it is what a
// sub-solve round costs): h and 1/(|d| + h) come from the hardware estimates v_rsq_f64 / v_rcp_f64 with one Newton step --
// an error in t only leaves a_pq' = O(eps a_pq) behind, far below what the next sweep removes -- while c gets the two Newton
// steps of full precision, so that c^2 + s^2 = c^2 (1 + t^2) = 1 to round-off and R stays orthogonal.
__device__ __forceinline__ void jacobi_rotation(double apq, double app, double aqq, double& c, double& s)
{
    c = 1.0; s = 0.0;
    const double apq2 = apq * apq;
    if (apq2 > 1e-300 && apq2 > 1e-36 * fabs(app * aqq)) {
        const double d = 0.5 * (aqq - app);
        const double x = d * d + apq2;
        double ih = __builtin_amdgcn_rsq(x);
        ih = ih * (1.5 - 0.5 * x * ih * ih);
        const double den = fabs(d) + x * ih;
        double rd = __builtin_amdgcn_rcp(den);
        rd = rd * (2.0 - den * rd);
        const double t = (d >= 0.0 ? apq : -apq) * rd;
        c = fast_rsqrt(1.0 + t * t);
        s = t * c;
    }
}

// One Jacobi round = ONE launch of jacobi_round_kernel with three kinds of workgroups:
//   [0, n_sub)   sub-solves of round `round_next`: R (row-major JS x JS) with R^T S R diagonal for every block pair -> rbuf_next;
//   the rest     updates of round `round` (rotations in rbuf): kind 0  A_next[P,Q] = R_P^T A_cur[P,Q] R_Q over the upper triangle
//                of pair blocks + mirror;  kind 1  the eigenvectors V[t,Q] <- V[t,Q] R_Q in place.
// The sub-problem (I',J') of the next round does not wait for the update.  Block b of it sits in exactly one pair p(b) of this
// round, so its off-diagonal block is one rotated tile, which the sub-solve workgroup computes itself from A_cur,
//      S'[I',J'] = ( R_p(I')^T . A_cur[p(I'), p(J')] . R_p(J') ) [half(I'), half(J')],
// and its two diagonal blocks are diagonal blocks of the matrices S_p = R_p^T S R_p that the sub-solves of this round ended with:
// they are handed on through dbuf (they differ from the blocks of A_next by the rounding of a different summation order, i.e.
// the rotation angles by a relative 1e-15 -- convergence is judged on A itself).  The dependency chain of a round is therefore
// one launch (~20 us of sub-solve latency) with the updates -- bound by the traffic of A and V through the Infinity Cache --
// beside it, where it used to be a sub-solve launch and an update launch back to back.  A is double-buffered for that (the
// updates write A_next while the sub-solves read A_cur).  The update needs ~7 workgroups per CU in flight to pull its traffic, so
// the sub-solve path is kept as light as the update path: S and R rotate in place (17 KB of LDS), one MFMA tile, <= 72 registers
// (a first fused version with double-buffered S and R and three FMA-rotated tiles, 34 KB / 100 registers, ran the launch in
// 39.5 us; the two kernels on two streams were slower than back to back because of the cross-stream waits).
struct UpdTask { int32_t mat, p, q, kind; };     // p: pair index (kind 0) or row tile (kind 1); q: pair index (both local to mat)
typedef double jd4 __attribute__((ext_vector_type(4)));

// Seat permutation of the round-robin tournament in "neighbours play" form: the JS indices sit in two rows of JB seats, seat
// 2t above seat 2t+1, and seat 2t always plays seat 2t+1.  After a round everybody except seat 0 moves one seat clockwise;
// after JS-1 rounds all pairs have met and everybody is back (period JS-1), so S and R end in their original order.
__device__ __forceinline__ int jacobi_next_seat(int s)
{
    if (s == 0) return 0;
    if (s == 1) return 2;
    if (s & 1) return s - 2;                 // lower row moves left
    return s == JS - 2 ? JS - 1 : s + 2;     // upper row moves right, the last one drops to the lower row
}
// "Cross" form, for a visit that only rotates the pairs BETWEEN the two blocks (the pairs inside a block are rotated once per
// sweep, in the full visits of its first round): block I sits in the upper row (even seats) and stays, block J in the lower row
// moves one seat left per round, and after JB rounds every (i, j) has met once and everybody is back.
__device__ __forceinline__ int jacobi_next_seat_cross(int s) { return (s & 1) ? (s + JS - 2) % JS : s; }
// seat of index j of the sub-problem ([block I; block J]) at the start (and at the end) of a visit
__device__ __forceinline__ int jacobi_seat_of(int j, int cross) { return cross ? (j < JB ? 2 * j : 2 * (j - JB) + 1) : j; }

// acc = R_P^T . M[rows, cols] . R_Q (two_sided) or M[rows, cols] . R_Q for one JS x JS tile, rows = blocks (IP, JP) of JB rows or
// JS consecutive rows from row0 (IP < 0), cols = blocks (IQ, JQ).  256 threads = 2 x 2 waves, each wave owns a 16 x 16 quarter
// of the tile as one v_mfma_f64_16x16x4 accumulator (fragment maps as in ggemm.hip: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// result col = l&15, row = (l>>4) + 4 reg); the tile goes through LDS (X, JS x JLD), R_Q and R_P^T go from global memory
// straight into MFMA fragments.  (The plain-FMA version of these products was LDS-bandwidth bound: the update ran at ~12 TF/s
// and took 70 % of the eigensolve at m = 2048.)  Leaves the once-rotated tile in X; ends without a barrier.
constexpr int JW = JS / 16;                    // the workgroup is a JW x JW grid of waves, one 16 x 16 MFMA block of the tile each
static_assert(SUB_THREADS == 64 * JW * JW, "one thread per pair of rotation pairs == one wave per 16 x 16 block of the tile");
__device__ __forceinline__ jd4 tile_rotate(double* X, const double* __restrict__ M, int npad, int IP, int JP, int row0, int IQ, int JQ,
                                           const double* __restrict__ Rq, const double* __restrict__ Rp, bool two_sided)
{
    constexpr int KG = JS / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / JW, wc = wave % JW;
    const int l15 = lane & 15, l4 = lane >> 4;
    auto cq = [&](int j) { return j < JB ? IQ * JB + j : JQ * JB + j - JB; };
    auto rp = [&](int i) { return IP >= 0 ? (i < JB ? IP * JB + i : JP * JB + i - JB) : row0 + i; };
    // fragments: B operand of X . R_Q is R_Q[k][16 wc + l15]; A operand of R_P^T . Y is R_P[k][16 wr + l15]
    double bq[KG], ap[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        bq[g] = Rq[(4 * g + l4) * JS + 16 * wc + l15];
        ap[g] = two_sided ? Rp[(4 * g + l4) * JS + 16 * wr + l15] : 0.0;
    }
    for (int e = tid; e < JS * JS; e += SUB_THREADS) {
        const int i = e / JS, j = e % JS;
        X[i * JLD + j] = M[(int64_t)rp(i) * npad + cq(j)];
    }
    __syncthreads();
    jd4 acc = (jd4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < KG; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(16 * wr + l15) * JLD + 4 * g + l4], bq[g], acc, 0, 0, 0);      // X . R_Q
    if (two_sided) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(16 * wr + l4 + 4 * r) * JLD + 16 * wc + l15] = acc[r];
        __syncthreads();
        acc = (jd4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KG; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[g], X[(4 * g + l4) * JLD + 16 * wc + l15], acc, 0, 0, 0);  // R_P^T . (X . R_Q)
    }
    return acc;
}

//   kind 0 (A, two-sided): block (P,Q), P <= Q, of the pair-block partition:  A_next[P,Q] = R_P^T . A_cur[P,Q] . R_Q, and its
//                          transpose is written to A_next[Q,P] -- every JS x JS block is written by exactly one workgroup, the
//                          column and the row update of the textbook formulation fuse into one pass over the upper
//                          triangle, and A stays symmetric to the last bit;
//   kind 1 (V, one-sided): rows [JS*t, JS*t+JS) x pair-block Q:      V[t,Q] <- V[t,Q] . R_Q
__device__ __forceinline__ void jacobi_update_body(double* X, const MatDesc* __restrict__ mats, const int32_t* __restrict__ pair_start, const UpdTask t,
                                                   double* __restrict__ buf, const double* __restrict__ rbuf, int round, int flip)
{
    const MatDesc m = mats[t.mat];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave / JW, wc = wave % JW, l15 = lane & 15, l4 = lane >> 4;
    int IQ, JQ, IP = -1, JP = -1;
    pair_blocks(m.nb, round, t.q, IQ, JQ);
    if (t.kind == 0) pair_blocks(m.nb, round, t.p, IP, JP);
    const double* Rq = rbuf + (int64_t)(pair_start[t.mat] + t.q) * JS * JS;
    const double* Rp = rbuf + (int64_t)(pair_start[t.mat] + t.p) * JS * JS;
    const double* M = buf + (t.kind == 0 ? (flip ? m.a2_off : m.a_off) : m.v_off);       // A: read the current buffer,
    double* Mo = buf + (t.kind == 0 ? (flip ? m.a_off : m.a2_off) : m.v_off);            //    write the other one; V: in place
    const jd4 acc = tile_rotate(X, M, m.npad, IP, JP, t.p * JS, IQ, JQ, Rq, Rp, t.kind == 0);
    auto cq = [&](int j) { return j < JB ? IQ * JB + j : JQ * JB + j - JB; };
    auto rp = [&](int i) { return t.kind == 0 ? (i < JB ? IP * JB + i : JP * JB + i - JB) : t.p * JS + i; };
#pragma unroll
    for (int r = 0; r < 4; ++r) Mo[(int64_t)rp(16 * wr + l4 + 4 * r) * m.npad + cq(16 * wc + l15)] = acc[r];
    if (t.kind == 0 && t.p != t.q) {                              // mirror: A[Q,P] = (A[P,Q])^T, staged through LDS for row-wise stores
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(16 * wr + l4 + 4 * r) * JLD + 16 * wc + l15] = acc[r];
        __syncthreads();
        for (int e = tid; e < JS * JS; e += SUB_THREADS) {
            const int i = e / JS, j = e % JS;                     // element (i, j) of the transposed block
            Mo[(int64_t)cq(i) * m.npad + rp(j)] = X[j * JLD + i];
        }
    }
}

// The sub-solve is a chain of JS-1 dependent rotation rounds; what a round costs is instructions issued (several sub-problems
// share a CU) plus two LDS round trips.  So the loop carries no index arithmetic at all: thread (k, l) always owns the 2 x 2
// block of seats {2k, 2k+1} x {2l, 2l+1} of S and rows 2k, 2k+1 x columns {2l, 2l+1} of R, reads it from fixed addresses and,
// after the barrier that also hands over the rotations, writes the rotated block to the fixed addresses of the seats its rows /
// columns move to (in place: every read of a round precedes that barrier, every write follows it); the JB rotations of a
// round are computed once, by the first JB lanes of wave 0 from the fixed pivot positions (2t, 2t+1), and handed over through LDS.
//
// flip: which buffer of A is current.  has_prev == 0: no rotation is pending (very first round): S' is A_cur itself and the
// launch holds sub-solves only.  dbuf / dbuf_next: per pair the two JB x JB diagonal blocks of the S it ended with.
__global__ void __launch_bounds__(SUB_THREADS)
jacobi_round_kernel(const MatDesc* __restrict__ mats, const PairRef* __restrict__ pairs, int n_sub, double* __restrict__ buf, int flip,
                    const double* __restrict__ rbuf, const double* __restrict__ dbuf, int round, int has_prev,
                    double* __restrict__ rbuf_next, double* __restrict__ dbuf_next, int round_next, int cross,
                    const UpdTask* __restrict__ tasks, const int32_t* __restrict__ pair_start)
{
    constexpr int BUFSZ = JS * JLD;
    constexpr bool STAGE_IN_R = (JB > 16);            // JB = 32: 33 KB per buffer -- the staging tile shares R's (two workgroups per CU)
    __shared__ double sh[(STAGE_IN_R ? 2 : 3) * BUFSZ];    // S, R and the staging tile of the off-diagonal block (JB = 16: 25 KB, the registers
                                                      // cap the kernel at 6 workgroups per CU anyway)
    __shared__ double rot_c[JB], rot_s[JB];
    if ((int)blockIdx.x >= n_sub) { jacobi_update_body(sh, mats, pair_start, tasks[blockIdx.x - n_sub], buf, rbuf, round, flip); return; }
    double* S = sh;
    double* R = sh + BUFSZ;
    const PairRef pr = pairs[blockIdx.x];
    const MatDesc m = mats[pr.mat];
    int I, J;
    pair_blocks(m.nb, round_next, pr.j, I, J);
    const int tid = threadIdx.x;
    const double* A = buf + (flip ? m.a2_off : m.a_off);
    if (!STAGE_IN_R) for (int e = tid; e < JS * JS; e += SUB_THREADS) R[(e / JS) * JLD + jacobi_seat_of(e % JS, cross)] = (e / JS == e % JS) ? 1.0 : 0.0;
    if (!has_prev) {
        for (int e = tid; e < JS * JS; e += SUB_THREADS) {
            const int i = e / JS, j = e % JS;
            const int gi = (i < JB ? I * JB + i : J * JB + i - JB), gj = (j < JB ? I * JB + j : J * JB + j - JB);
            S[jacobi_seat_of(i, cross) * JLD + jacobi_seat_of(j, cross)] = A[(int64_t)gi * m.npad + gj];
        }
    } else {
        // blocks (I, J) of the coming round, as they are once this round's rotations are applied
        int jp0, hp0, jp1, hp1, PI0, PJ0, PI1, PJ1;
        block_seat(m.nb, round, I, jp0, hp0);
        block_seat(m.nb, round, J, jp1, hp1);
        pair_blocks(m.nb, round, jp0, PI0, PJ0);
        pair_blocks(m.nb, round, jp1, PI1, PJ1);
        const int64_t g0 = pair_start[pr.mat] + jp0, g1 = pair_start[pr.mat] + jp1;
        // diagonal blocks: loads issued before the tile rotation, stored after it
        double dv[2 * JB * JB / SUB_THREADS];
#pragma unroll
        for (int u = 0; u < 2 * JB * JB / SUB_THREADS; ++u)
            dv[u] = dbuf[(u == 0 ? g0 * 2 + hp0 : g1 * 2 + hp1) * (JB * JB) + tid];
        const jd4 acc = tile_rotate(STAGE_IN_R ? R : sh + 2 * BUFSZ, A, m.npad, PI0, PJ0, 0, PI1, PJ1, rbuf + g1 * JS * JS, rbuf + g0 * JS * JS, true);
        const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
        // the block (half hp0 of the rows, half hp1 of the columns) of the rotated tile is a JW/2 x JW/2 group of waves
        if ((wave / JW) / (JW / 2) == hp0 && (wave % JW) / (JW / 2) == hp1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int si = jacobi_seat_of(16 * ((wave / JW) % (JW / 2)) + l4 + 4 * r, cross), sj = jacobi_seat_of(JB + 16 * ((wave % JW) % (JW / 2)) + l15, cross);
                S[si * JLD + sj] = acc[r]; S[sj * JLD + si] = acc[r];
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * JB * JB / SUB_THREADS; ++u) S[jacobi_seat_of(u * JB + tid / JB, cross) * JLD + jacobi_seat_of(u * JB + tid % JB, cross)] = dv[u];
    }
    if (STAGE_IN_R) {
        __syncthreads();                                // the staging tile is read no more
        for (int e = tid; e < JS * JS; e += SUB_THREADS) R[(e / JS) * JLD + jacobi_seat_of(e % JS, cross)] = (e / JS == e % JS) ? 1.0 : 0.0;
    }
    __syncthreads();
    {
        // (no convergence test per sub-problem: a launch lasts as long as its slowest sub-solve, rotations of negligible elements are
        //  skipped one by one in jacobi_rotation, and the test itself -- a reduction and two barriers -- cost 0.7 us of every round)
        {
            const int k = tid / JB, l = tid % JB;
            // loop-invariant LDS offsets (elements): the block this thread reads, and where its two rows / columns go
            const int r0 = 2 * k, r1 = 2 * k + 1, c0 = 2 * l, c1 = 2 * l + 1;
            const int nr0 = cross ? jacobi_next_seat_cross(r0) : jacobi_next_seat(r0), nr1 = cross ? jacobi_next_seat_cross(r1) : jacobi_next_seat(r1);
            const int nc0 = cross ? jacobi_next_seat_cross(c0) : jacobi_next_seat(c0), nc1 = cross ? jacobi_next_seat_cross(c1) : jacobi_next_seat(c1);
            const int nrounds = cross ? JB : JS - 1;
            const int in00 = r0 * JLD + c0, in01 = r0 * JLD + c1, in10 = r1 * JLD + c0, in11 = r1 * JLD + c1;
            const int so00 = nr0 * JLD + nc0, so01 = nr0 * JLD + nc1, so10 = nr1 * JLD + nc0, so11 = nr1 * JLD + nc1;     // S: rows and columns move
            const int ro00 = r0 * JLD + nc0, ro01 = r0 * JLD + nc1, ro10 = r1 * JLD + nc0, ro11 = r1 * JLD + nc1;         // R: only columns move
            const int piv = (2 * tid) * JLD + 2 * tid;                                                                  // lane t < JB: pivot block (2t, 2t+1)
#pragma unroll 1
            for (int rr = 0; rr < nrounds; ++rr) {
                if (tid < JB) {                                            // wave 0, JB lanes: one rotation each
                    double c, sn;
                    jacobi_rotation(S[piv + 1], S[piv], S[piv + JLD + 1], c, sn);
                    rot_c[tid] = c; rot_s[tid] = sn;
                }
                const double b00 = S[in00], b01 = S[in01], b10 = S[in10], b11 = S[in11];
                const double q00 = R[in00], q01 = R[in01], q10 = R[in10], q11 = R[in11];
                __syncthreads();
                const double ck = rot_c[k], sk = rot_s[k], cl = rot_c[l], sl = rot_s[l];
                const double t00 = cl * b00 - sl * b01, t01 = sl * b00 + cl * b01;
                const double t10 = cl * b10 - sl * b11, t11 = sl * b10 + cl * b11;
                S[so00] = ck * t00 - sk * t10; S[so01] = ck * t01 - sk * t11;
                S[so10] = sk * t00 + ck * t10; S[so11] = sk * t01 + ck * t11;
                R[ro00] = cl * q00 - sl * q01; R[ro01] = sl * q00 + cl * q01;
                R[ro10] = cl * q10 - sl * q11; R[ro11] = sl * q10 + cl * q11;
                __syncthreads();
            }
        }
    }
    double* Rout = rbuf_next + (int64_t)blockIdx.x * JS * JS;
    for (int e = tid; e < JS * JS; e += SUB_THREADS) Rout[e] = R[(e / JS) * JLD + jacobi_seat_of(e % JS, cross)];
    double* Dout = dbuf_next + (int64_t)blockIdx.x * 2 * JB * JB;
    for (int e = tid; e < 2 * JB * JB; e += SUB_THREADS) {
        const int h = e / (JB * JB), r = (e % (JB * JB)) / JB, c = e % JB;
        Dout[e] = S[jacobi_seat_of(h * JB + r, cross) * JLD + jacobi_seat_of(h * JB + c, cross)];
    }
}

// per matrix and block of the grid: out[(mat*NORM_BLOCKS + b)*2] = partial sum of squares off the diagonal, [..+1] = on it
constexpr int NORM_BLOCKS = 32;
__global__ void __launch_bounds__(256) offnorm_kernel(const MatDesc* __restrict__ mats, const double* __restrict__ buf, double* __restrict__ out, int flip)
{
    __shared__ double r0[4], r1[4];
    MatDesc m = mats[blockIdx.y];
    if (flip) m.a_off = m.a2_off;
    double off = 0.0, dg = 0.0;
    const int64_t tot = (int64_t)m.npad * m.npad;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += 256 * NORM_BLOCKS) {
        const int i = (int)(e / m.npad), j = (int)(e % m.npad);
        const double v = buf[m.a_off + e];
        if (i == j) dg += v * v; else off += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); dg += __shfl_down(dg, o, 64); }
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = off; r1[threadIdx.x >> 6] = dg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[(blockIdx.y * NORM_BLOCKS + blockIdx.x) * 2] = r0[0] + r0[1] + r0[2] + r0[3];
        out[(blockIdx.y * NORM_BLOCKS + blockIdx.x) * 2 + 1] = r1[0] + r1[1] + r1[2] + r1[3];
    }
}

__global__ void diag_kernel(const MatDesc* __restrict__ mats, const double* buf, double* out, const int64_t* __restrict__ out_off)
{
    const MatDesc m = mats[blockIdx.x];
    for (int i = threadIdx.x; i < m.npad; i += blockDim.x) out[out_off[blockIdx.x] + i] = buf[m.a_off + (int64_t)i * m.npad + i];
}

// Thousands of plane rotations leave the columns of V orthogonal to ~1e-15 but their norms drift by ~1e-13:
// renormalise every column once at the end (64 columns per workgroup, coalesced along the row).
__global__ void __launch_bounds__(256) normalize_columns_kernel(const MatDesc* __restrict__ mats, double* __restrict__ buf)
{
    __shared__ double part[4][64];
    const MatDesc m = mats[blockIdx.y];
    const int c0 = blockIdx.x * 64;
    if (c0 >= m.npad) return;
    const int col = c0 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    double* V = buf + m.v_off;
    double s = 0.0;
    if (col < m.npad) for (int i = rg; i < m.npad; i += 4) { const double v = V[(int64_t)i * m.npad + col]; s += v * v; }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    const double tot = part[0][threadIdx.x & 63] + part[1][threadIdx.x & 63] + part[2][threadIdx.x & 63] + part[3][threadIdx.x & 63];
    const double inv = tot > 0.0 ? 1.0 / sqrt(tot) : 0.0;
    if (col < m.npad) for (int i = rg; i < m.npad; i += 4) V[(int64_t)i * m.npad + col] *= inv;
}

// out[c] = sum_i W[i*ld + c]^2 for c < ncols  (Rayleigh quotients: lambda_c = |Psi^T u_c|^2)
struct ColNormTask { int64_t w_off, out_off; int32_t nrows, ncols, ld, pad; };
__global__ void __launch_bounds__(256) colnorm_kernel(const ColNormTask* __restrict__ tasks, const double* buf, double* out)
{
    __shared__ double part[4][64];
    const ColNormTask t = tasks[blockIdx.y];
    const int c0 = blockIdx.x * 64;
    if (c0 >= t.ncols) return;
    const int col = c0 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    double s = 0.0;
    if (col < t.ncols) for (int i = rg; i < t.nrows; i += 4) { const double v = buf[t.w_off + (int64_t)i * t.ld + col]; s += v * v; }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && col < t.ncols) out[t.out_off + col] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// Preconditioner input: B = [ P^T A P | I ] (n x 2n) with P the permutation that sorts the diagonal of A downwards
__global__ void __launch_bounds__(256) qr_gather_kernel(const MatDesc* __restrict__ mats, const HqrMat* __restrict__ qm, const int32_t* __restrict__ perm,
                                                         const int64_t* __restrict__ perm_off, double* __restrict__ buf)
{
    const MatDesc m = mats[blockIdx.y];
    const HqrMat q = qm[blockIdx.y];
    const int n = q.n;
    const int32_t* pm = perm + perm_off[blockIdx.y];
    const int64_t tot = (int64_t)n * 2 * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / (2 * n)), j = (int)(e % (2 * n));
        buf[q.b_off + e] = j < n ? buf[m.a_off + (int64_t)pm[i] * m.npad + pm[j]] : (j - n == i ? 1.0 : 0.0);
    }
}

// Preconditioner output: E (n x n, rows = the new basis in the original index order):  E[r][perm[j]] = Q^T[r][j]
__global__ void __launch_bounds__(256) qr_scatter_kernel(const HqrMat* __restrict__ qm, const int32_t* __restrict__ perm, const int64_t* __restrict__ perm_off,
                                                          const int64_t* __restrict__ e_off, double* __restrict__ buf)
{
    const HqrMat q = qm[blockIdx.y];
    const int n = q.n;
    const int32_t* pm = perm + perm_off[blockIdx.y];
    const int64_t tot = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / n), j = (int)(e % n);
        buf[e_off[blockIdx.y] + (int64_t)r * n + pm[j]] = buf[q.b_off + (int64_t)r * 2 * n + n + j];
    }
}

// dst[r*ld + i] = V[i*npad + perm[r]]
__global__ void gather_vec_kernel(const double* __restrict__ V, int npad, int n, const int32_t* __restrict__ perm, int count, double* __restrict__ dst, int64_t ld)
{
    const int r = blockIdx.y;
    if (r >= count) return;
    const int col = perm[r];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[(int64_t)r * ld + i] = V[(int64_t)i * npad + col];
}

// the same for up to GATHER_MAX (matrix, destination) pairs in one launch; the descriptors travel in the kernel-argument segment
constexpr int GATHER_MAX = 48;
struct GatherOne { const double* V; const int32_t* perm; double* dst; int64_t ld; int32_t npad, n, count, pad; };
struct GatherArgs { GatherOne t[GATHER_MAX]; };
__global__ void gather_vec_batch_kernel(const GatherArgs a)
{
    const GatherOne g = a.t[blockIdx.z];
    for (int r = blockIdx.y; r < g.count; r += gridDim.y) {
        const int col = g.perm[r];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) g.dst[(int64_t)r * g.ld + i] = g.V[(int64_t)i * g.npad + col];
    }
}

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

struct dmrgx_rdm {
    int32_t nblocks = 0;
    std::vector<MatDesc> mats;                 // index 2*k + side
    DevBuf d_tables;                    // the set-up tables in one upload (d_mats is a view into it)
    DevBuf buf, d_mats, d_perm;
    std::vector<std::vector<double>> eig;      // per matrix: eigenvalues, descending
    std::vector<std::vector<int32_t>> perm;    // per matrix: column of V for the r-th largest eigenvalue
    std::vector<int64_t> perm_off;
    std::vector<uint8_t> selected;             // per matrix: built and diagonalised by this rank (dmrgx_rdm_create_subset)
    int32_t sweeps = 0;
    int32_t solver = 0;                        // 0: tridiagonalisation + divide and conquer (symeig.hip), 1: block Jacobi
    SymEigReport symeig;                       // what the direct solver did (solver == 0)
    // ---- second phase (dmrgx_rdm_select; direct solver only): the eigenvectors of the kept states are formed once the caller has cut ----
    SymEigDeferred deferred;                   // pending: the spectra are final, the eigenvectors are not formed yet
    std::vector<int32_t> have;                 // per matrix: eigenvectors (and Rayleigh quotients) exist for the `have` largest eigenvalues
    const double* psi = nullptr;               // the state the matrices were built from: must stay alive until the selection
    std::vector<int64_t> off, diag_off;        // offsets of the KronBlocks in psi; of every matrix in the per-position arrays
    std::vector<int32_t> nl, nr;               // sector sizes of every KronBlock
    int64_t psiT_off = 0, w_base = 0, rq_base = 0, ew_base = 0, diag_base = 0, dtot = 0;
    // the Rayleigh quotients of the selected states travel to the host behind the launches that follow them and are compared with the
    // solver's own eigenvalues when the object is destroyed (or verified): no synchronisation of their own
    double* rq_host = nullptr;                 // pinned, from pinned_take
    size_t rq_cap = 0;
    std::vector<int32_t> check_cols;
    hipEvent_t check_event = nullptr;          // recorded behind the read-back: the verification waits for IT, not for the stream
    bool check_pending = false;
    ~dmrgx_rdm();
};

// Pinned host blocks for those read-backs, recycled per host thread (hipHostMalloc / hipHostFree per truncation would cost more than the
// synchronisation they replace).
namespace {
struct PinnedCache { std::vector<std::pair<double*, size_t>> free; };      // (never freed: a thread-exit hipHostFree can run after the runtime's own teardown)
PinnedCache& pinned_cache() { static thread_local PinnedCache c; return c; }
double* pinned_take(size_t count, size_t* cap)
{
    auto& f = pinned_cache().free;
    for (size_t i = 0; i < f.size(); ++i) if (f[i].second >= count) { double* p = f[i].first; *cap = f[i].second; f.erase(f.begin() + (std::ptrdiff_t)i); return p; }
    double* p = nullptr;
    const size_t n = std::max<size_t>(count + count / 4, 4096);
    if (hipHostMalloc((void**)&p, n * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap = n;
    return p;
}
}  // namespace
dmrgx_rdm::~dmrgx_rdm()
{
    if (rq_host) pinned_cache().free.push_back({rq_host, rq_cap});
    if (check_event) (void)hipEventDestroy(check_event);
}

static dmrgx_status rdm_rayleigh(dmrgx_rdm* P, const std::vector<int32_t>& cols, bool check_against_direct, hipStream_t st);

static dmrgx_status rdm_create_impl(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                    const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                    const double* const* v0_rows, const uint8_t* side_mask, void* stream, dmrgx_rdm** out);

extern "C" dmrgx_status dmrgx_rdm_create(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                         const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                         void* stream, dmrgx_rdm** out)
{
    return dmrgx_rdm_create_warm(left, right, nblocks, block_il, block_ir, psi_dev, nullptr, stream, out);
}

extern "C" dmrgx_status dmrgx_rdm_create_warm(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                              const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                              const double* const* v0_rows, void* stream, dmrgx_rdm** out)
{
    return rdm_create_impl(left, right, nblocks, block_il, block_ir, psi_dev, v0_rows, nullptr, stream, out);
}

extern "C" dmrgx_status dmrgx_rdm_create_subset(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                                const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                                const uint8_t* side_mask, void* stream, dmrgx_rdm** out)
{
    if (!side_mask) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_create_subset: null side_mask");
    return rdm_create_impl(left, right, nblocks, block_il, block_ir, psi_dev, nullptr, side_mask, stream, out);
}

static dmrgx_status rdm_create_impl(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                    const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                    const double* const* v0_rows, const uint8_t* side_mask, void* stream, dmrgx_rdm** out)
{
    hipStream_t st = (hipStream_t)stream;
    if (!left || !right || !block_il || !block_ir || !psi_dev || !out || nblocks <= 0) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_create: bad argument");
    *out = nullptr;
    std::unique_ptr<dmrgx_rdm> P(new (std::nothrow) dmrgx_rdm());
    if (!P) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
    P->nblocks = nblocks;
    static const bool stage_trace = getenv("DMRGX_RDM_TRACE") != nullptr;      // developer aid: wall time of every stage (synchronising)
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* name) {
        if (!stage_trace) return;
        (void)hipStreamSynchronize(st);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[rdm] stage %-10s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // ---- solver: tridiagonalisation + divide and conquer (symeig.hip, round 3) unless the environment asks for the block-Jacobi
    //      iteration of rounds 1-2 (DMRGX_RDM_SOLVER=jacobi; kept for comparison and for orders above SYMEIG_MAX_N)
    static const bool want_dc = !(getenv("DMRGX_RDM_SOLVER") && std::string(getenv("DMRGX_RDM_SOLVER")) == "jacobi");
    bool use_dc = want_dc;                            // (a caller's warm-start bases are then only a hint, as with the QR preconditioner)
    for (int32_t k = 0; k < nblocks && use_dc; ++k) {
        if (block_il[k] < 0 || block_il[k] >= left->nsec || block_ir[k] < 0 || block_ir[k] >= right->nsec) break;       // reported below
        if (std::max(left->size[block_il[k]], right->size[block_ir[k]]) > SYMEIG_MAX_N) use_dc = false;
    }
    // ---- layout -------------------------------------------------------------------------------------------
    std::vector<int64_t> off(nblocks + 1, 0);
    int64_t total = 0;
    for (int32_t k = 0; k < nblocks; ++k) {
        const int32_t il = block_il[k], ir = block_ir[k];
        if (il < 0 || il >= left->nsec || ir < 0 || ir >= right->nsec) DMRGX_FAIL(DMRGX_ERR_OUTOFRANGE, "rdm_create: KronBlock %d out of range", k);
        const int32_t nl = left->size[il], nr = right->size[ir];
        off[k + 1] = off[k] + (int64_t)nl * nr;
        for (int side = 0; side < 2; ++side) {
            MatDesc m;
            const bool sel = !side_mask || ((side_mask[k] >> side) & 1);
            P->selected.push_back(sel ? 1 : 0);
            m.n = !sel ? 0 : side == 0 ? nl : nr;             // a matrix left to another rank is an empty matrix here
            m.npad = ((m.n + JS - 1) / JS) * JS;
            m.nb = m.npad / JB;
            m.a_off = total; total += (int64_t)m.npad * m.npad;
            m.a2_off = total; total += use_dc ? 0 : (int64_t)m.npad * m.npad;      // second buffer of A: block Jacobi only
            m.v_off = total; total += (int64_t)m.npad * m.npad;
            m.pad = 0;
            P->mats.push_back(m);
        }
    }
    const int64_t N = off[nblocks];
    const int64_t psiT_off = total; total += N;
    const int nm = (int)P->mats.size();
    std::vector<PairRef> pairs;
    std::vector<UpdTask> tiles;              // the A updates of a round, then its V updates
    std::vector<int32_t> pair_start(nm);
    int max_nb = 2;
    for (int mi = 0; mi < nm; ++mi) {
        const MatDesc& m = P->mats[mi];
        max_nb = std::max(max_nb, m.nb);
        pair_start[mi] = (int32_t)pairs.size();
        const int np = m.nb / 2;
        for (int j = 0; j < np; ++j) pairs.push_back(PairRef{mi, j});
        for (int p = 0; p < np; ++p) for (int q = p; q < np; ++q) tiles.push_back(UpdTask{mi, p, q, 0});     // upper triangle; the kernel mirrors
    }
    for (int mi = 0; mi < nm; ++mi) {
        const MatDesc& m = P->mats[mi];
        for (int t = 0; t < m.npad / JS; ++t) for (int q = 0; q < m.nb / 2; ++q) tiles.push_back(UpdTask{mi, t, q, 1});
    }
    const int64_t rbuf_off = total, rbuf_len = (int64_t)pairs.size() * JS * JS; total += 2 * rbuf_len;    // two slots: round r and r+1
    const int64_t dbuf_off = total, dbuf_len = (int64_t)pairs.size() * 2 * JB * JB; total += 2 * dbuf_len;  // diagonal blocks handed from round to round
    const int64_t norm_off = total; total += 2 * nm * NORM_BLOCKS;
    std::vector<int64_t> diag_off(nm);
    int64_t dtot = 0;
    for (int mi = 0; mi < nm; ++mi) { diag_off[mi] = dtot; dtot += P->mats[mi].npad; }
    const int64_t diag_base = total; total += dtot;
    const int64_t rq_base = total; total += dtot;            // Rayleigh quotients, same indexing as the diagonals
    const int64_t w_base = total; total += 2 * N;            // W = Psi^T V_L (n_R x n_L) and Psi V_R (n_L x n_R) per block
    P->solver = use_dc ? 0 : 1;
    const int64_t ew_base = total; total += use_dc ? dtot : 0;   // eigenvalues as the direct solver returns them (ascending)
    // warm start: per matrix with a previous eigenbasis E (rows), W = E A and E^T (n x n each)
    // QR preconditioner of the Jacobi path: its basis replaces the caller's v0_rows, which then only remain a hint
    const bool use_qr = !use_dc;
    std::vector<const double*> warm_src(nm, nullptr);
    std::vector<HqrMat> qmats(nm, HqrMat{0, 0, 0, 0, 0});
    std::vector<int64_t> qe_off(nm, 0), qperm_off(nm, 0);
    int64_t qperm_tot = 0;
    bool any_qr = false;
    for (int mi = 0; mi < nm; ++mi) {
        const int64_t n = P->mats[mi].n;
        if (use_qr && n >= 2 && n <= HQR_MAX_N) {
            qmats[mi].n = (int32_t)n;
            qmats[mi].b_off = total; total += 2 * n * n;
            qmats[mi].v_off = total; total += 32 * n;
            qmats[mi].t_off = total; total += 32 * 32;
            qe_off[mi] = total; total += n * n;
            qperm_off[mi] = qperm_tot; qperm_tot += n;
            any_qr = true;
        } else if (!use_qr && !use_dc && v0_rows && v0_rows[mi]) warm_src[mi] = v0_rows[mi];
    }
    std::vector<int64_t> warm_w(nm, -1), warm_et(nm, -1);
    for (int mi = 0; mi < nm; ++mi) if (warm_src[mi] || qmats[mi].n) { const int64_t nn = (int64_t)P->mats[mi].n * P->mats[mi].n; warm_w[mi] = total; total += nn; warm_et[mi] = total; total += nn; }
    DMRGX_CHK(P->buf.alloc((size_t)total * sizeof(double)));
    double* buf = P->buf.as<double>();
    // ---- tables of the set-up (matrix descriptors, Jacobi pair lists, transposition tiles, the Gram GEMMs) in ONE upload ----------
    DevBuf d_pairs, d_tiles, d_doff, d_pstart;          // views into P->d_tables
    for (auto& v : diag_off) v += diag_base;
    std::vector<TrTile> tt;
    for (int32_t k = 0; k < nblocks; ++k) {
        const int32_t nl = left->size[block_il[k]], nr = right->size[block_ir[k]];
        if (!P->selected[2 * k] && !P->selected[2 * k + 1]) continue;
        for (int ti = 0; ti < (nl + 31) / 32; ++ti) for (int tj = 0; tj < (nr + 31) / 32; ++tj)
            tt.push_back(TrTile{off[k], psiT_off + off[k], nl, nr, ti, tj});
    }
    std::vector<GProd> gprods;
    std::vector<GGroup> ggroups;
    std::vector<GTile> gt, gb;
    for (int32_t k = 0; k < nblocks; ++k) {
        const int32_t nl = left->size[block_il[k]], nr = right->size[block_ir[k]];
        const double* Psi = psi_dev + off[k];
        const double* PsiT = buf + psiT_off + off[k];
        const MatDesc& mL = P->mats[2 * k];
        const MatDesc& mR = P->mats[2 * k + 1];
        if (P->selected[2 * k]) {
            gprods.push_back(GProd{Psi, PsiT, nr, nl, nr, GPROD_GEMM, 1.0});       // rho_L = Psi Psi^T  (:1733)
            ggroups.push_back(GGroup{buf + mL.a_off, mL.npad, nl, nl, (int32_t)gprods.size() - 1, (int32_t)gprods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)ggroups.size() - 1, nl, nl, (nr + GG_BK - 1) / GG_BK);
        }
        if (P->selected[2 * k + 1]) {
            gprods.push_back(GProd{PsiT, Psi, nl, nr, nl, GPROD_GEMM, 1.0});       // rho_R = Psi^T Psi  (:1734)
            ggroups.push_back(GGroup{buf + mR.a_off, mR.npad, nr, nr, (int32_t)gprods.size() - 1, (int32_t)gprods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)ggroups.size() - 1, nr, nr, (nl + GG_BK - 1) / GG_BK);
        }
    }
    if (gprods.empty()) gprods.push_back(GProd{nullptr, nullptr, 0, 0, 0, GPROD_GEMM, 0.0});
    if (ggroups.empty()) ggroups.push_back(GGroup{nullptr, 0, 0, 0, 0, 0, 0, 0});
    ggemm_schedule(gt, ggroups); ggemm_schedule(gb, ggroups, 2);
    size_t o_tt = 0, o_gp = 0, o_gg = 0, o_gt = 0, o_gb = 0;
    {
        PackedUpload pk;
        const size_t o_m = pk.add(P->mats), o_p = pk.add(pairs), o_t = pk.add(tiles), o_ps = pk.add(pair_start), o_d = pk.add(diag_off);
        o_tt = pk.add(tt); o_gp = pk.add(gprods); o_gg = pk.add(ggroups); o_gt = pk.add(gt); o_gb = pk.add(gb);
        DMRGX_CHK(pk.upload(P->d_tables, st));
        PackedUpload::view<MatDesc>(P->d_mats, P->d_tables, o_m, P->mats.size());
        PackedUpload::view<std::decay<decltype(pairs[0])>::type>(d_pairs, P->d_tables, o_p, pairs.size());
        PackedUpload::view<std::decay<decltype(tiles[0])>::type>(d_tiles, P->d_tables, o_t, tiles.size());
        PackedUpload::view<std::decay<decltype(pair_start[0])>::type>(d_pstart, P->d_tables, o_ps, pair_start.size());
        PackedUpload::view<int64_t>(d_doff, P->d_tables, o_d, diag_off.size());
    }
    const MatDesc* dm = P->d_mats.as<MatDesc>();
    hipLaunchKernelGGL(rdm_init_kernel, dim3(64, nm), dim3(256), 0, st, dm, buf);
    DMRGX_HIP(hipGetLastError());

    stage("layout");
    // ---- Psi^T and the 2*nblocks Gram matrices in one grouped GEMM launch -------------------------------------
    {
        // transpose reads psi (caller memory) and writes the arena: pass distinct base pointers
        if (!tt.empty()) hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)tt.size()), dim3(256), 0, st, (const TrTile*)packed_at<TrTile>(P->d_tables, o_tt), psi_dev, buf);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(ggemm_launch(packed_at<GTile>(P->d_tables, o_gb), packed_at<GGroup>(P->d_tables, o_gg), packed_at<GProd>(P->d_tables, o_gp), (int32_t)gb.size(), st, 1));
        DMRGX_CHK(ggemm_launch(packed_at<GTile>(P->d_tables, o_gt), packed_at<GGroup>(P->d_tables, o_gg), packed_at<GProd>(P->d_tables, o_gp), (int32_t)gt.size(), st, 0));
    }

    stage("gram");
    // ---- warm start: A <- E A E^T, V <- E^T for matrices whose previous eigenbasis E (eigenvectors as rows) is supplied.
    //      In a settled DMRG sweep the basis of the previous visit nearly diagonalises the new density matrix (measured:
    //      off^2/total^2 ~ 1e-5 instead of 0.5), which saves the first ~3 of ~10 Jacobi sweeps; the rest is the linearly
    //      converging tail of near-degenerate small eigenvalues.  Any orthogonal E is valid.
    if (any_qr) {
        // diagonal of every matrix -> host, sort downwards, gather [P^T A P | I], factor, scatter Q^T back to the original order
        std::vector<double> hd((size_t)dtot);
        hipLaunchKernelGGL(diag_kernel, dim3(nm), dim3(256), 0, st, dm, (const double*)buf, buf, d_doff.as<int64_t>());
        DMRGX_HIP(hipGetLastError());
        DMRGX_HIP(hipMemcpyAsync(hd.data(), buf + diag_base, hd.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        std::vector<int32_t> qperm((size_t)qperm_tot);
        int max_n = 0;
        for (int mi = 0; mi < nm; ++mi) {
            const int n = qmats[mi].n;
            if (!n) continue;
            max_n = std::max(max_n, n);
            const double* d = hd.data() + (diag_off[mi] - diag_base);
            int32_t* pm = qperm.data() + qperm_off[mi];
            std::iota(pm, pm + n, 0);
            std::stable_sort(pm, pm + n, [&](int32_t a, int32_t b) { return d[a] > d[b]; });
            warm_src[mi] = buf + qe_off[mi];
        }
        DevBuf d_qm, d_qperm, d_qpoff, d_qeoff;
        DMRGX_CHK(upload(d_qm, qmats, st)); DMRGX_CHK(upload(d_qperm, qperm, st)); DMRGX_CHK(upload(d_qpoff, qperm_off, st)); DMRGX_CHK(upload(d_qeoff, qe_off, st));
        const unsigned gx = (unsigned)std::min<int64_t>(512, ((int64_t)max_n * 2 * max_n + 255) / 256);
        hipLaunchKernelGGL(qr_gather_kernel, dim3(gx, nm), dim3(256), 0, st, dm, d_qm.as<HqrMat>(), d_qperm.as<int32_t>(), d_qpoff.as<int64_t>(), buf);
        DMRGX_HIP(hipGetLastError());
        DMRGX_CHK(hqr_batched(qmats, d_qm.as<HqrMat>(), buf, st));
        hipLaunchKernelGGL(qr_scatter_kernel, dim3(gx, nm), dim3(256), 0, st, d_qm.as<HqrMat>(), d_qperm.as<int32_t>(), d_qpoff.as<int64_t>(), d_qeoff.as<int64_t>(), buf);
        DMRGX_HIP(hipGetLastError());
    }
    stage("qr");
    std::vector<SymEigMat> sm;
    if (use_dc) {
        for (int mi = 0; mi < nm; ++mi) {
            const MatDesc& m = P->mats[mi];
            if (m.n > 0) sm.push_back(SymEigMat{m.n, m.npad, m.npad, 0, buf + m.a_off, buf + m.v_off, buf + ew_base + (diag_off[mi] - diag_base)});
        }
        DMRGX_CHK(symeig_batched(sm, st, &P->symeig, &P->deferred));
        stage("symeig");
    }
    bool any_warm = false;
    for (int mi = 0; mi < nm; ++mi) any_warm = any_warm || warm_src[mi];
    if (any_warm) {
        std::vector<TrTile> tt;
        std::vector<GProd> p1, p2;
        std::vector<GGroup> g1, g2;
        std::vector<GTile> t1, t1b, t2, t2b;
        for (int mi = 0; mi < nm; ++mi) {
            if (!warm_src[mi]) continue;
            const MatDesc& m = P->mats[mi];
            const int32_t n = m.n;
            if (n == 0) continue;
            const double* E = warm_src[mi];
            double* Wt = buf + warm_w[mi];
            double* ET = buf + warm_et[mi];
            for (int ti = 0; ti < (n + 31) / 32; ++ti) for (int tj = 0; tj < (n + 31) / 32; ++tj)
                tt.push_back(TrTile{(int64_t)(E - buf), warm_et[mi], n, n, ti, tj});     // source addressed relative to the arena base
            p1.push_back(GProd{E, buf + m.a_off, n, m.npad, n, GPROD_GEMM, 1.0});          // W = E A
            g1.push_back(GGroup{Wt, n, n, n, (int32_t)p1.size() - 1, (int32_t)p1.size(), 0, 0});
            ggemm_append_tiles_mixed(t1b, t1, (int32_t)g1.size() - 1, n, n, (n + GG_BK - 1) / GG_BK);
            p2.push_back(GProd{Wt, ET, n, n, n, GPROD_GEMM, 1.0});                          // A = W E^T
            g2.push_back(GGroup{buf + m.a_off, m.npad, n, n, (int32_t)p2.size() - 1, (int32_t)p2.size(), 0, 0});
            ggemm_append_tiles_mixed(t2b, t2, (int32_t)g2.size() - 1, n, n, (n + GG_BK - 1) / GG_BK);
        }
        if (!tt.empty()) {
            DevBuf d_tt, dp1, dg1, dt1, db1, dp2, dg2, dt2, db2;
            DMRGX_CHK(upload(d_tt, tt, st));
            hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)tt.size()), dim3(256), 0, st, d_tt.as<TrTile>(), (const double*)buf, buf);
            DMRGX_HIP(hipGetLastError());
            ggemm_schedule(t1, g1); ggemm_schedule(t1b, g1, 2); ggemm_schedule(t2, g2); ggemm_schedule(t2b, g2, 2);
            DMRGX_CHK(upload(dp1, p1, st)); DMRGX_CHK(upload(dg1, g1, st)); DMRGX_CHK(upload(dt1, t1, st)); DMRGX_CHK(upload(db1, t1b, st));
            DMRGX_CHK(upload(dp2, p2, st)); DMRGX_CHK(upload(dg2, g2, st)); DMRGX_CHK(upload(dt2, t2, st)); DMRGX_CHK(upload(db2, t2b, st));
            DMRGX_CHK(ggemm_launch(db1.as<GTile>(), dg1.as<GGroup>(), dp1.as<GProd>(), (int32_t)t1b.size(), st, 1));
            DMRGX_CHK(ggemm_launch(dt1.as<GTile>(), dg1.as<GGroup>(), dp1.as<GProd>(), (int32_t)t1.size(), st, 0));
            DMRGX_CHK(ggemm_launch(db2.as<GTile>(), dg2.as<GGroup>(), dp2.as<GProd>(), (int32_t)t2b.size(), st, 1));
            DMRGX_CHK(ggemm_launch(dt2.as<GTile>(), dg2.as<GGroup>(), dp2.as<GProd>(), (int32_t)t2.size(), st, 0));
            for (int mi = 0; mi < nm; ++mi) {
                if (!warm_src[mi] || P->mats[mi].n == 0) continue;
                const MatDesc& m = P->mats[mi];
                DMRGX_HIP(hipMemcpy2DAsync(buf + m.v_off, (size_t)m.npad * sizeof(double), buf + warm_et[mi], (size_t)m.n * sizeof(double),
                                           (size_t)m.n * sizeof(double), (size_t)m.n, hipMemcpyDeviceToDevice, st));
            }
        }
    }

    stage("transform");
    // ---- batched block Jacobi ------------------------------------------------------------------------------------
    std::vector<double> norms((size_t)2 * nm * NORM_BLOCKS);
    const int rounds = std::max(1, max_nb - 1);
    int sweep = 0, slot = 0, flip = 0;
    bool have_rot = false;                   // rbuf[slot] / dbuf[slot] hold the rotations of the round about to be applied
    for (; sweep < 30 && !use_dc; ++sweep) {
        hipLaunchKernelGGL(offnorm_kernel, dim3(NORM_BLOCKS, nm), dim3(256), 0, st, dm, buf, buf + norm_off, flip);
        DMRGX_HIP(hipGetLastError());
        DMRGX_HIP(hipMemcpyAsync(norms.data(), buf + norm_off, norms.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        bool conv = true;
        static const bool trace = getenv("DMRGX_RDM_TRACE") != nullptr;      // developer aid: convergence history on stderr
        double worst = 0.0;
        // off <= 1e-12 ||A||_F: V is a product of rotations (orthogonal to round-off whatever the convergence) and the
        // eigenvalues are re-evaluated as Rayleigh quotients below, whose error is O(off^2)
        for (int mi = 0; mi < nm; ++mi) {
            double off2 = 0.0, dg2 = 0.0;
            for (int b = 0; b < NORM_BLOCKS; ++b) { off2 += norms[(size_t)(mi * NORM_BLOCKS + b) * 2]; dg2 += norms[(size_t)(mi * NORM_BLOCKS + b) * 2 + 1]; }
            if (off2 > 1e-24 * (off2 + dg2)) conv = false;
            worst = std::max(worst, off2 / std::max(off2 + dg2, 1e-300));
        }
        if (trace) fprintf(stderr, "[rdm] sweep %d: max off^2/total^2 = %.3e\n", sweep, worst);
        if (conv) break;
        if (pairs.empty()) continue;
        for (int r = 0; r < rounds; ++r) {
            double* rcur = buf + rbuf_off + slot * rbuf_len;
            double* rnext = buf + rbuf_off + (slot ^ 1) * rbuf_len;
            double* dcur = buf + dbuf_off + slot * dbuf_len;
            double* dnext = buf + dbuf_off + (slot ^ 1) * dbuf_len;
            if (!have_rot) {                 // very first round: sub-solves alone, straight from A
                hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned)pairs.size()), dim3(SUB_THREADS), 0, st, dm, d_pairs.as<PairRef>(), (int)pairs.size(), buf, flip,
                                   (const double*)rcur, (const double*)dcur, 0, 0, rcur, dcur, r, 0, d_tiles.as<UpdTask>(), d_pstart.as<int32_t>());
                DMRGX_HIP(hipGetLastError());
                have_rot = true;
            }
            // apply round r (A_cur -> A_next, V in place) and, beside it, solve the sub-problems of the round after it
            const int r_next = (r + 1 == rounds) ? 0 : r + 1;
            const int cross_next = r_next != 0 ? 1 : 0;      // the pairs inside a block: once per sweep, in its first round
            hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned)(pairs.size() + tiles.size())), dim3(SUB_THREADS), 0, st, dm, d_pairs.as<PairRef>(), (int)pairs.size(), buf, flip,
                               (const double*)rcur, (const double*)dcur, r, 1, rnext, dnext, r_next, cross_next, d_tiles.as<UpdTask>(), d_pstart.as<int32_t>());
            DMRGX_HIP(hipGetLastError());
            slot ^= 1; flip ^= 1;
        }
    }
    P->sweeps = sweep;
    if (sweep >= 30) DMRGX_FAIL(DMRGX_ERR_NOTCONV, "rdm_create: block Jacobi did not converge in 30 sweeps");

    if (!use_dc) {           // (the direct solver's eigenvectors are products of Householder and GEMM factors: no drift to remove)
        int max_npad = JS;
        for (auto& m : P->mats) max_npad = std::max(max_npad, m.npad);
        hipLaunchKernelGGL(normalize_columns_kernel, dim3((max_npad + 63) / 64, nm), dim3(256), 0, st, dm, buf);
        DMRGX_HIP(hipGetLastError());
    }
    stage("jacobi");
    // ---- what the second phase needs ------------------------------------------------------------------------------------------
    P->psi = psi_dev; P->off = off; P->diag_off = diag_off;
    P->psiT_off = psiT_off; P->w_base = w_base; P->rq_base = rq_base; P->ew_base = ew_base; P->diag_base = diag_base; P->dtot = dtot;
    for (int32_t k = 0; k < nblocks; ++k) { P->nl.push_back(left->size[block_il[k]]); P->nr.push_back(right->size[block_ir[k]]); }
    P->eig.resize(nm); P->perm.resize(nm); P->perm_off.resize(nm); P->have.assign((size_t)nm, 0);
    if (use_dc) {
        // Direct solver, first phase: the spectra are final (secular equations of the divide and conquer), ascending per matrix; the
        // eigenvectors are formed by dmrgx_rdm_select for the states the caller keeps (or for all of them at the first request).
        std::vector<double> ew((size_t)dtot);
        if (!ew.empty()) DMRGX_HIP(hipMemcpyAsync(ew.data(), buf + ew_base, ew.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        if (const int32_t bad = symeig_deferred_timed_out(P->deferred)) {
            // the persistent tridiagonalisation did not get its partners: everything queued behind it ran on garbage.  Once more, by launches
            // (the density matrices themselves were only read).
            symeig_note_timeout(bad);
            P->deferred.dbuf.release(); P->deferred.ibuf.release(); P->deferred.pending = false;
            DMRGX_CHK(symeig_batched(sm, st, &P->symeig, &P->deferred));
            P->symeig.timed_out = bad;
            if (!ew.empty()) DMRGX_HIP(hipMemcpyAsync(ew.data(), buf + ew_base, ew.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            DMRGX_HIP(hipStreamSynchronize(st));
            if (symeig_deferred_timed_out(P->deferred)) DMRGX_FAIL(DMRGX_ERR_INTERNAL, "rdm_create: the tridiagonalisation by launches reported a persistent-kernel status");
        }
        std::vector<int32_t> allperm;
        for (int mi = 0; mi < nm; ++mi) {
            const MatDesc& m = P->mats[mi];
            const double* a = ew.data() + (diag_off[mi] - diag_base);
            P->eig[mi].resize(m.n); P->perm[mi].resize(m.n);
            for (int32_t r = 0; r < m.n; ++r) {
                const double v = a[m.n - 1 - r];
                if (!(v == v)) DMRGX_FAIL(DMRGX_ERR_NOTCONV, "rdm_create: density matrix %d (n = %d): the direct solver returned NaN", mi, m.n);
                P->eig[mi][r] = v; P->perm[mi][r] = m.n - 1 - r;      // r-th largest eigenvalue = column n - 1 - r
            }
            P->perm_off[mi] = (int64_t)allperm.size();
            allperm.insert(allperm.end(), P->perm[mi].begin(), P->perm[mi].end());
        }
        DMRGX_CHK(upload(P->d_perm, allperm, st));
        stage("spectra");
        *out = P.release();
        return DMRGX_OK;
    }
    // ---- block Jacobi: eigenvalues as Rayleigh quotients of the renormalised eigenvectors, all columns ------------------------------
    {
        std::vector<int32_t> cols((size_t)nm);
        for (int mi = 0; mi < nm; ++mi) cols[(size_t)mi] = P->mats[mi].n;
        DMRGX_CHK(rdm_rayleigh(P.get(), cols, false, st));
    }
    stage("rayleigh");
    *out = P.release();
    return DMRGX_OK;
}

// Eigenvalues as Rayleigh quotients of the finished eigenvectors: lambda = |Psi^T u|^2 (rho_L) / |Psi v|^2 (rho_R).  Relative accuracy ~eps
// instead of the c*n*eps*||rho|| of a rotated diagonal / a secular root, which matters for TruncErr = 1 - sum of kept eigenvalues
// (include/DMRGBlockContainer.hpp:1872-1875).  cols[mi] columns of matrix mi are evaluated: for the block-Jacobi solver all of them, in
// whatever order the iteration left them (sorted here); for the direct solver the LAST cols[mi] columns -- the kept states -- whose
// D&C eigenvalues they replace (and are checked against).
static dmrgx_status rdm_rayleigh(dmrgx_rdm* P, const std::vector<int32_t>& cols, bool direct, hipStream_t st)
{
    const int nm = (int)P->mats.size();
    const int32_t nblocks = P->nblocks;
    double* buf = P->buf.as<double>();
    const int64_t dtot = P->dtot;
    std::vector<double> rq((size_t)dtot, 0.0);
    {
        std::vector<GProd> prods;
        std::vector<GGroup> groups;
        std::vector<GTile> gt, gb;
        std::vector<ColNormTask> cn;
        for (int32_t k = 0; k < nblocks; ++k) {
            const int32_t nl = P->nl[(size_t)k], nr = P->nr[(size_t)k];
            const double* Psi = P->psi + P->off[(size_t)k];
            const double* PsiT = buf + P->psiT_off + P->off[(size_t)k];
            const MatDesc& mL = P->mats[2 * k];
            const MatDesc& mR = P->mats[2 * k + 1];
            double* WL = buf + P->w_base + 2 * P->off[(size_t)k];                     // n_R x (columns of V_L)
            double* WR = WL + (int64_t)nl * nr;                                        // n_L x (columns of V_R)
            const int32_t cL = P->selected[2 * k] ? cols[(size_t)(2 * k)] : 0, cR = P->selected[2 * k + 1] ? cols[(size_t)(2 * k + 1)] : 0;
            if (cL > 0) {
                const int32_t c0 = nl - cL;
                prods.push_back(GProd{PsiT, buf + mL.v_off + c0, nl, mL.npad, nl, GPROD_GEMM, 1.0});
                groups.push_back(GGroup{WL, cL, nr, cL, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
                ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nr, cL, (nl + GG_BK - 1) / GG_BK);
                cn.push_back(ColNormTask{P->w_base + 2 * P->off[(size_t)k], P->rq_base + (P->diag_off[2 * k] - P->diag_base) + c0, nr, cL, cL, 0});
            }
            if (cR > 0) {
                const int32_t c0 = nr - cR;
                prods.push_back(GProd{Psi, buf + mR.v_off + c0, nr, mR.npad, nr, GPROD_GEMM, 1.0});
                groups.push_back(GGroup{WR, cR, nl, cR, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
                ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nl, cR, (nr + GG_BK - 1) / GG_BK);
                cn.push_back(ColNormTask{P->w_base + 2 * P->off[(size_t)k] + (int64_t)nl * nr, P->rq_base + (P->diag_off[2 * k + 1] - P->diag_base) + c0, nl, cR, cR, 0});
            }
        }
        if (prods.empty()) return DMRGX_OK;                 // (nothing selected on this rank)
        ggemm_schedule(gt, groups); ggemm_schedule(gb, groups, 2);
        DevBuf dtab;
        PackedUpload pk;
        const size_t o_p = pk.add(prods), o_g = pk.add(groups), o_t = pk.add(gt), o_b = pk.add(gb), o_c = pk.add(cn);
        DMRGX_CHK(pk.upload(dtab, st));
        DMRGX_HIP(zero_async(buf + P->rq_base, (size_t)dtot * sizeof(double), st));
        DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtab, o_b), packed_at<GGroup>(dtab, o_g), packed_at<GProd>(dtab, o_p), (int32_t)gb.size(), st, 1));
        DMRGX_CHK(ggemm_launch(packed_at<GTile>(dtab, o_t), packed_at<GGroup>(dtab, o_g), packed_at<GProd>(dtab, o_p), (int32_t)gt.size(), st, 0));
        int maxc = 1;
        for (auto& c : cn) maxc = std::max(maxc, c.ncols);
        hipLaunchKernelGGL(colnorm_kernel, dim3((maxc + 63) / 64, (unsigned)cn.size()), dim3(256), 0, st, (const ColNormTask*)packed_at<ColNormTask>(dtab, o_c), buf, buf);
        DMRGX_HIP(hipGetLastError());
        if (direct) {
            // Direct solver: the spectrum the caller has (and cut on) is the solver's own; the Rayleigh quotients of the finished columns are
            // the second, independent route to the kept eigenvalues -- compared when the object is destroyed, after the caller's own
            // synchronisation (dmrgx_rdm_destroy returns DMRGX_ERR_NOTCONV on a mismatch).  Nothing waits here.
            if (!P->rq_host) P->rq_host = pinned_take((size_t)dtot, &P->rq_cap);
            if (!P->rq_host) DMRGX_FAIL(DMRGX_ERR_MEM, "rdm_select: no pinned host memory for the verification read-back");
            DMRGX_HIP(hipMemcpyAsync(P->rq_host, buf + P->rq_base, (size_t)dtot * sizeof(double), hipMemcpyDeviceToHost, st));
            if (!P->check_event) DMRGX_HIP(hipEventCreateWithFlags(&P->check_event, hipEventDisableTiming));
            DMRGX_HIP(hipEventRecord(P->check_event, st));
            P->check_cols = cols; P->check_pending = true;
            for (int mi = 0; mi < nm; ++mi) if (P->selected[mi] && P->mats[mi].n > 0) P->have[(size_t)mi] = cols[(size_t)mi];
            return DMRGX_OK;
        }
        DMRGX_HIP(hipMemcpyAsync(rq.data(), buf + P->rq_base, rq.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
    }
    // ---- block Jacobi: sort descending on the host, remember the column of V for each rank ----------------------------------------------
    std::vector<int32_t> allperm;
    for (int mi = 0; mi < nm; ++mi) {
        const MatDesc& m = P->mats[mi];
        // Padding indices (>= n) are decoupled (zero off-diagonals, rotations with a_pq == 0 are skipped), so the real
        // eigenvectors are exactly the columns [0, n) of V and the padding columns stay unit vectors.
        const double* q = rq.data() + (P->diag_off[mi] - P->diag_base);
        std::vector<int32_t> real(m.n);
        std::iota(real.begin(), real.end(), 0);
        std::stable_sort(real.begin(), real.end(), [&](int32_t a, int32_t b) { return q[a] > q[b]; });
        P->perm[mi] = real;
        P->eig[mi].resize(m.n);
        for (int32_t r = 0; r < m.n; ++r) P->eig[mi][r] = q[real[r]];
        P->perm_off[mi] = (int64_t)allperm.size();
        allperm.insert(allperm.end(), real.begin(), real.end());
        P->have[(size_t)mi] = m.n;
    }
    DMRGX_CHK(upload(P->d_perm, allperm, st));
    return DMRGX_OK;
}

// counts == NULL: every eigenvector of every matrix
static dmrgx_status rdm_select_impl(dmrgx_rdm* R, const int32_t* counts, hipStream_t st)
{
    if (!R->deferred.pending) return DMRGX_OK;
    const int nm = (int)R->mats.size();
    std::vector<int32_t> cols((size_t)nm, 0), keep;
    for (int mi = 0; mi < nm; ++mi) {
        const int32_t n = R->mats[mi].n;
        if (n <= 0) continue;
        const int32_t c = counts ? counts[mi] : n;
        if (c < 0 || c > n) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_select: matrix (block %d, side %d) of order %d cannot keep %d states", mi / 2, mi % 2, n, c);
        cols[(size_t)mi] = c;
        keep.push_back(c);                               // (the direct solver saw the matrices of order > 0, in this order)
    }
    DMRGX_CHK(symeig_finish(R->deferred, keep, st));
    return rdm_rayleigh(R, cols, true, st);
}

extern "C" dmrgx_status dmrgx_rdm_select(dmrgx_rdm* R, const int32_t* counts, void* stream)
{
    if (!R) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_select: bad argument");
    if (!R->deferred.pending && R->solver == 0 && counts) {
        // a second selection can only narrow what the first one formed
        for (size_t mi = 0; mi < R->mats.size(); ++mi)
            if (R->selected[mi] && R->mats[mi].n > 0 && counts[mi] > R->have[mi])
                DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_select: the eigenvectors of matrix (block %d, side %d) were formed for %d states, %d asked", (int)mi / 2, (int)mi % 2, R->have[mi], counts[mi]);
    }
    return rdm_select_impl(R, counts, (hipStream_t)stream);
}

extern "C" dmrgx_status dmrgx_rdm_eigenvalues(const dmrgx_rdm* R, int32_t side, int32_t k, double* host_out)
{
    if (!R || !host_out || side < 0 || side > 1 || k < 0 || k >= R->nblocks) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvalues: bad argument");
    if (!R->selected[2 * k + side]) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvalues: the density matrix (block %d, side %d) was left to another rank", k, side);
    const auto& e = R->eig[2 * k + side];
    std::copy(e.begin(), e.end(), host_out);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_eigenvectors(const dmrgx_rdm* R, int32_t side, int32_t k, int32_t count, double* dst_dev, int64_t ld, void* stream)
{
    if (!R || side < 0 || side > 1 || k < 0 || k >= R->nblocks || count < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: bad argument");
    const int mi = 2 * k + side;
    if (!R->selected[mi]) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: the density matrix (block %d, side %d) was left to another rank", k, side);
    const MatDesc& m = R->mats[mi];
    if (count > m.n || (count > 0 && (!dst_dev || ld < m.n))) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: count %d > n %d or bad destination", count, m.n);
    if (count == 0) return DMRGX_OK;
    // (a caller that never selected gets every eigenvector, formed at its first request: the one-phase behaviour)
    if (R->deferred.pending) DMRGX_CHK(rdm_select_impl(const_cast<dmrgx_rdm*>(R), nullptr, (hipStream_t)stream));
    if (count > R->have[(size_t)mi]) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: %d eigenvectors asked of (block %d, side %d), dmrgx_rdm_select formed %d", count, k, side, R->have[(size_t)mi]);
    hipLaunchKernelGGL(gather_vec_kernel, dim3((m.n + 255) / 256, count), dim3(256), 0, (hipStream_t)stream,
                       R->buf.as<double>() + m.v_off, m.npad, m.n, R->d_perm.as<int32_t>() + R->perm_off[mi], count, dst_dev, ld);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_eigenvectors_batch(const dmrgx_rdm* R, int32_t ntasks, const dmrgx_rdm_vec_task* tasks, void* stream)
{
    if (!R || ntasks < 0 || (ntasks > 0 && !tasks)) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors_batch: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int32_t t = 0; t < ntasks; ++t) {
        const dmrgx_rdm_vec_task& q = tasks[t];
        if (q.side < 0 || q.side > 1 || q.k < 0 || q.k >= R->nblocks || q.count < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors_batch: task %d: bad argument", t);
        const int mi = 2 * q.k + q.side;
        if (!R->selected[mi]) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors_batch: the density matrix (block %d, side %d) was left to another rank", q.k, q.side);
        const MatDesc& m = R->mats[mi];
        if (q.count > m.n || (q.count > 0 && (!q.dst_dev || q.ld < m.n))) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors_batch: task %d: count %d > n %d or bad destination", t, q.count, m.n);
    }
    if (R->deferred.pending) DMRGX_CHK(rdm_select_impl(const_cast<dmrgx_rdm*>(R), nullptr, st));
    for (int32_t t0 = 0; t0 < ntasks; t0 += GATHER_MAX) {
        GatherArgs a;
        int na = 0, nmax = 1, cmax = 1;
        for (int32_t t = t0; t < std::min(ntasks, t0 + GATHER_MAX); ++t) {
            const dmrgx_rdm_vec_task& q = tasks[t];
            const int mi = 2 * q.k + q.side;
            const MatDesc& m = R->mats[mi];
            if (q.count == 0) continue;
            if (q.count > R->have[(size_t)mi]) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors_batch: %d eigenvectors asked of (block %d, side %d), dmrgx_rdm_select formed %d", q.count, q.k, q.side, R->have[(size_t)mi]);
            a.t[na++] = GatherOne{R->buf.as<double>() + m.v_off, R->d_perm.as<int32_t>() + R->perm_off[mi], q.dst_dev, q.ld, m.npad, m.n, q.count, 0};
            nmax = std::max(nmax, m.n); cmax = std::max(cmax, q.count);
        }
        if (na == 0) continue;
        hipLaunchKernelGGL(gather_vec_batch_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)std::min(cmax, 4096), (unsigned)na), dim3(256), 0, st, a);
        DMRGX_HIP(hipGetLastError());
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_info(const dmrgx_rdm* R, dmrgx_rdm_report* out)
{
    if (!R || !out) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_info: bad argument");
    memset(out, 0, sizeof(*out));
    out->n_sweeps = R->sweeps;
    out->solver = R->solver;
    out->trid_persistent_matrices = R->symeig.persistent_matrices;
    out->trid_launch_matrices = R->symeig.launch_matrices;
    out->max_workgroups_per_matrix = R->symeig.max_workgroups_per_matrix;
    out->merge_levels = R->symeig.merge_levels;
    out->wy_blocks_max = R->symeig.wy_blocks_max;
    out->timed_out = R->symeig.timed_out;
    symeig_process_state(&out->process_timeouts, &out->persistent_off);
    return DMRGX_OK;
}

// Two independent routes to every kept eigenvalue: the secular equations of the divide and conquer (what the caller cut on) and the Rayleigh
// quotient |Psi^T u|^2 of the finished eigenvector.  They agree to round-off; a failed leaf or secular solve (a wrong root) or a broken
// eigenvector shows here instead of as a wrong truncation further down -- the counterpart of the reference's "all eigenpairs
// converged" check (include/DMRGBlockContainer.hpp:1987).
static dmrgx_status rdm_verify(dmrgx_rdm* R)
{
    if (!R->check_pending) return DMRGX_OK;
    R->check_pending = false;
    DMRGX_HIP(hipEventSynchronize(R->check_event));      // (only the read-back: work queued on the stream since is not waited for)
    for (size_t mi = 0; mi < R->mats.size(); ++mi) {
        const MatDesc& m = R->mats[mi];
        const int32_t c = R->selected[mi] ? R->check_cols[mi] : 0;
        if (m.n <= 0 || c <= 0) continue;
        const double* b = R->rq_host + (R->diag_off[mi] - R->diag_base);
        const double scale = std::fabs(R->eig[mi][0]);
        double worst = 0.0;
        bool bad = false;
        for (int32_t r = 0; r < c; ++r) { const double d = std::fabs(R->eig[mi][r] - b[m.n - 1 - r]); if (!(d <= worst)) worst = d; if (!(d == d)) bad = true; }
        if (bad || !(worst <= 1e-8 * scale + 1e-300))
            DMRGX_FAIL(DMRGX_ERR_NOTCONV, "rdm: density matrix (block %d, side %d, n = %d): the direct solver's eigenvalues and the Rayleigh quotients of its vectors differ by %.3e (largest eigenvalue %.3e)",
                       (int)mi / 2, (int)mi % 2, m.n, worst, scale);
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_destroy(dmrgx_rdm* R)
{
    if (!R) return DMRGX_OK;
    // The verification of the selected eigenpairs is read here: it waits for the event behind its read-back only -- a caller that destroys the
    // object one step later (the engine does) never waits at all; the blocks go back to the pool and are recycled in stream order (pool.hip)
    const dmrgx_status rc = rdm_verify(R);
    delete R;
    return rc;
}
