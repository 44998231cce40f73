// K3/K4: reduced density matrices of the superblock ground state and their full eigendecomposition.
//
// Replaces the rank-0 loop of GetTruncation (reference include/DMRGBlockContainer.hpp:1715-1775): for every
// KronBlock k, Psi = n_L x n_R row-major slice of psi (:1731), rho_L = Psi Psi^T, rho_R = Psi^T Psi (:1733-1734),
// then EigRDM_BlockDiag = all eigenpairs of each block by LAPACK (:1962-2003).  Here all 2*nblocks matrices are
// built by ONE launch of the grouped MFMA-f64 GEMM and diagonalised TOGETHER on the device by a batched two-sided
// block-Jacobi method (block size 32): per round every matrix contributes nb/2 disjoint block pairs; a 64 x 64
// sub-problem is solved by scalar Jacobi in LDS, then the rotation is applied to the block columns of A and V
// and to the block rows of A.  Sorting, the global m-cut and the stable re-sort by sector (:1795,1850-1853) stay
// on the host (caller), which fetches spectra with dmrgx_rdm_eigenvalues and asks for the kept eigenvectors with
// dmrgx_rdm_eigenvectors (== FillRotation_BlockDiag, :2006-2057).
#include "ggemm.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

namespace dmrgx {
namespace {

constexpr int JB = 32, JS = 2 * JB;          // block size, sub-problem size
constexpr int JLD = JS + 1;

struct MatDesc { int64_t a_off, v_off; int32_t n, npad, nb, pad; };
struct PairRef { int32_t mat, j; };
struct TileRef { int32_t pair, tile; };

__device__ __forceinline__ void pair_blocks(int nb, int r, int j, int& I, int& J)
{
    const int m1 = nb - 1;
    if (m1 == 0) { I = 0; J = 0; return; }
    const int rr = r % m1;
    if (j == 0) { I = m1; J = rr; }
    else { I = (rr + j) % m1; J = (rr - j + m1) % m1; }
    if (I > J) { const int t = I; I = J; J = t; }
}

// A <- 0 with -1 on the padding diagonal; V <- identity
__global__ void rdm_init_kernel(const MatDesc* __restrict__ mats, double* __restrict__ buf)
{
    const MatDesc m = mats[blockIdx.y];
    const int64_t tot = (int64_t)m.npad * m.npad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / m.npad), j = (int)(e % m.npad);
        buf[m.v_off + e] = (i == j) ? 1.0 : 0.0;
        if (i >= m.n || j >= m.n) buf[m.a_off + e] = (i == j) ? -1.0 : 0.0;
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

// Solve the 64 x 64 symmetric sub-problem of block pair (I,J) by cyclic Jacobi in LDS; R (row-major 64x64) such
// that R^T S R is diagonal is written to rbuf[pair].
__global__ void __launch_bounds__(256)
jacobi_sub_kernel(const MatDesc* __restrict__ mats, const PairRef* __restrict__ pairs, double* __restrict__ buf, double* __restrict__ rbuf, int round)
{
    __shared__ double S[JS * JLD], R[JS * JLD];
    __shared__ double cs[JB], sn[JB];
    __shared__ int pp[JB], qq[JB];
    __shared__ double red0[4], red1[4];
    const PairRef pr = pairs[blockIdx.x];
    const MatDesc m = mats[pr.mat];
    int I, J;
    pair_blocks(m.nb, round, pr.j, I, J);
    const int tid = threadIdx.x;
    double* A = buf + m.a_off;
    for (int e = tid; e < JS * JS; e += 256) {
        const int i = e / JS, j = e % JS;
        const int gi = (i < JB ? I * JB + i : J * JB + i - JB), gj = (j < JB ? I * JB + j : J * JB + j - JB);
        S[i * JLD + j] = A[(int64_t)gi * m.npad + gj];
        R[i * JLD + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 12; ++sweep) {
        // convergence: off-diagonal mass relative to the diagonal (wave-uniform decision)
        double off = 0.0, dg = 0.0;
        for (int e = tid; e < JS * JS; e += 256) {
            const int i = e / JS, j = e % JS;
            const double v = S[i * JLD + j];
            if (i == j) dg += v * v; else off += v * v;
        }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); dg += __shfl_down(dg, o, 64); }
        if ((tid & 63) == 0) { red0[tid >> 6] = off; red1[tid >> 6] = dg; }
        __syncthreads();
        const double offt = red0[0] + red0[1] + red0[2] + red0[3], dgt = red1[0] + red1[1] + red1[2] + red1[3];
        __syncthreads();
        if (offt <= 1e-32 * dgt) break;
        for (int rr = 0; rr < JS - 1; ++rr) {
            if (tid < JB) {
                int p, q;
                if (tid == 0) { p = JS - 1; q = rr; }
                else { p = (rr + tid) % (JS - 1); q = (rr - tid + (JS - 1)) % (JS - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = S[p * JLD + q], app = S[p * JLD + p], aqq = S[q * JLD + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                }
                cs[tid] = c; sn[tid] = s; pp[tid] = p; qq[tid] = q;
            }
            __syncthreads();
            // columns of S and R:  [x_p, x_q] <- [c x_p - s x_q, s x_p + c x_q]
            for (int e = tid; e < JB * JS; e += 256) {
                const int k = e / JS, i = e % JS;
                const int p = pp[k], q = qq[k];
                const double c = cs[k], s = sn[k];
                const double sp = S[i * JLD + p], sq = S[i * JLD + q];
                S[i * JLD + p] = c * sp - s * sq; S[i * JLD + q] = s * sp + c * sq;
                const double rp = R[i * JLD + p], rq = R[i * JLD + q];
                R[i * JLD + p] = c * rp - s * rq; R[i * JLD + q] = s * rp + c * rq;
            }
            __syncthreads();
            // rows of S
            for (int e = tid; e < JB * JS; e += 256) {
                const int k = e / JS, j = e % JS;
                const int p = pp[k], q = qq[k];
                const double c = cs[k], s = sn[k];
                const double sp = S[p * JLD + j], sq = S[q * JLD + j];
                S[p * JLD + j] = c * sp - s * sq; S[q * JLD + j] = s * sp + c * sq;
            }
            __syncthreads();
        }
    }
    double* Rout = rbuf + (int64_t)blockIdx.x * JS * JS;
    for (int e = tid; e < JS * JS; e += 256) Rout[e] = R[(e / JS) * JLD + (e % JS)];
}

// out(64x64) = L(64x64) * M(64x64), all in LDS; thread computes a 4x4 block
__device__ __forceinline__ void lds_mm64(const double* L, const double* M, double acc[4][4], int ty, int tx)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k = 0; k < JS; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = L[(4 * ty + i) * JLD + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = M[k * JLD + 4 * tx + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
}

// mode 0: X[rows of tile, cols(I,J)] <- X * R   for X = A and X = V     (grid.y = 2 selects A / V)
// mode 1: A[rows(I,J), cols of tile] <- R^T * A
__global__ void __launch_bounds__(256)
jacobi_update_kernel(const MatDesc* __restrict__ mats, const PairRef* __restrict__ pairs, const TileRef* __restrict__ tiles,
                     double* __restrict__ buf, const double* __restrict__ rbuf, int round, int mode)
{
    __shared__ double L[JS * JLD], M[JS * JLD];
    const TileRef tr = tiles[blockIdx.x];
    const PairRef pr = pairs[tr.pair];
    const MatDesc m = mats[pr.mat];
    int I, J;
    pair_blocks(m.nb, round, pr.j, I, J);
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const double* Rg = rbuf + (int64_t)tr.pair * JS * JS;
    double* X = buf + ((mode == 0 && blockIdx.y == 1) ? m.v_off : m.a_off);
    const int t0 = tr.tile * JS;
    auto gidx = [&](int i) { return i < JB ? I * JB + i : J * JB + i - JB; };
    if (mode == 0) {
        for (int e = tid; e < JS * JS; e += 256) {
            const int i = e / JS, j = e % JS;
            L[i * JLD + j] = X[(int64_t)(t0 + i) * m.npad + gidx(j)];
            M[i * JLD + j] = Rg[e];
        }
    } else {
        for (int e = tid; e < JS * JS; e += 256) {
            const int i = e / JS, j = e % JS;
            L[j * JLD + i] = Rg[e];                                   // L = R^T
            M[i * JLD + j] = X[(int64_t)gidx(i) * m.npad + t0 + j];
        }
    }
    __syncthreads();
    double acc[4][4];
    lds_mm64(L, M, acc, ty, tx);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * ty + i, c = 4 * tx + j;
            if (mode == 0) X[(int64_t)(t0 + r) * m.npad + gidx(c)] = acc[i][j];
            else X[(int64_t)gidx(r) * m.npad + t0 + c] = acc[i][j];
        }
}

// per matrix: out[2*mat] = sum of squares off the diagonal, out[2*mat+1] = on the diagonal
__global__ void __launch_bounds__(256) offnorm_kernel(const MatDesc* __restrict__ mats, const double* __restrict__ buf, double* __restrict__ out)
{
    __shared__ double r0[4], r1[4];
    const MatDesc m = mats[blockIdx.x];
    double off = 0.0, dg = 0.0;
    const int64_t tot = (int64_t)m.npad * m.npad;
    for (int64_t e = threadIdx.x; e < tot; e += 256) {
        const int i = (int)(e / m.npad), j = (int)(e % m.npad);
        const double v = buf[m.a_off + e];
        if (i == j) dg += v * v; else off += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); dg += __shfl_down(dg, o, 64); }
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = off; r1[threadIdx.x >> 6] = dg; }
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = r0[0] + r0[1] + r0[2] + r0[3]; out[2 * blockIdx.x + 1] = r1[0] + r1[1] + r1[2] + r1[3]; }
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
    for (int i = rg; i < m.npad; i += 4) { const double v = V[(int64_t)i * m.npad + col]; s += v * v; }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    const double tot = part[0][threadIdx.x & 63] + part[1][threadIdx.x & 63] + part[2][threadIdx.x & 63] + part[3][threadIdx.x & 63];
    const double inv = tot > 0.0 ? 1.0 / sqrt(tot) : 0.0;
    for (int i = rg; i < m.npad; i += 4) V[(int64_t)i * m.npad + col] *= inv;
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

// dst[r*ld + i] = V[i*npad + perm[r]]
__global__ void gather_vec_kernel(const double* __restrict__ V, int npad, int n, const int32_t* __restrict__ perm, int count, double* __restrict__ dst, int64_t ld)
{
    const int r = blockIdx.y;
    if (r >= count) return;
    const int col = perm[r];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[(int64_t)r * ld + i] = V[(int64_t)i * npad + col];
}

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

struct dmrgx_rdm {
    int32_t nblocks = 0;
    std::vector<MatDesc> mats;                 // index 2*k + side
    DevBuf buf, d_mats, d_perm;
    std::vector<std::vector<double>> eig;      // per matrix: eigenvalues, descending
    std::vector<std::vector<int32_t>> perm;    // per matrix: column of V for the r-th largest eigenvalue
    std::vector<int64_t> perm_off;
    int32_t sweeps = 0;
};

extern "C" dmrgx_status dmrgx_rdm_create(const dmrgx_sectors* left, const dmrgx_sectors* right, int32_t nblocks,
                                         const int32_t* block_il, const int32_t* block_ir, const double* psi_dev,
                                         void* stream, dmrgx_rdm** out)
{
    hipStream_t st = (hipStream_t)stream;
    if (!left || !right || !block_il || !block_ir || !psi_dev || !out || nblocks <= 0) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_create: bad argument");
    *out = nullptr;
    std::unique_ptr<dmrgx_rdm> P(new (std::nothrow) dmrgx_rdm());
    if (!P) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
    P->nblocks = nblocks;
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
            m.n = side == 0 ? nl : nr;
            m.npad = ((m.n + JS - 1) / JS) * JS;
            m.nb = m.npad / JB;
            m.a_off = total; total += (int64_t)m.npad * m.npad;
            m.v_off = total; total += (int64_t)m.npad * m.npad;
            m.pad = 0;
            P->mats.push_back(m);
        }
    }
    const int64_t N = off[nblocks];
    const int64_t psiT_off = total; total += N;
    const int nm = (int)P->mats.size();
    std::vector<PairRef> pairs;
    std::vector<TileRef> tiles;
    int max_nb = 2;
    for (int mi = 0; mi < nm; ++mi) {
        const MatDesc& m = P->mats[mi];
        max_nb = std::max(max_nb, m.nb);
        for (int j = 0; j < m.nb / 2; ++j) {
            pairs.push_back(PairRef{mi, j});
            for (int t = 0; t < m.npad / JS; ++t) tiles.push_back(TileRef{(int32_t)pairs.size() - 1, t});
        }
    }
    const int64_t rbuf_off = total; total += (int64_t)pairs.size() * JS * JS;
    const int64_t norm_off = total; total += 2 * nm;
    std::vector<int64_t> diag_off(nm);
    int64_t dtot = 0;
    for (int mi = 0; mi < nm; ++mi) { diag_off[mi] = dtot; dtot += P->mats[mi].npad; }
    const int64_t diag_base = total; total += dtot;
    const int64_t rq_base = total; total += dtot;            // Rayleigh quotients, same indexing as the diagonals
    const int64_t w_base = total; total += 2 * N;            // W = Psi^T V_L (n_R x n_L) and Psi V_R (n_L x n_R) per block
    DMRGX_CHK(P->buf.alloc((size_t)total * sizeof(double)));
    double* buf = P->buf.as<double>();
    DevBuf d_pairs, d_tiles, d_doff;
    DMRGX_CHK(upload(P->d_mats, P->mats, st));
    DMRGX_CHK(upload(d_pairs, pairs, st));
    DMRGX_CHK(upload(d_tiles, tiles, st));
    for (auto& v : diag_off) v += diag_base;
    DMRGX_CHK(upload(d_doff, diag_off, st));
    const MatDesc* dm = P->d_mats.as<MatDesc>();
    hipLaunchKernelGGL(rdm_init_kernel, dim3(64, nm), dim3(256), 0, st, dm, buf);
    DMRGX_HIP(hipGetLastError());

    // ---- Psi^T and the 2*nblocks Gram matrices in one grouped GEMM launch -------------------------------------
    {
        std::vector<TrTile> tt;
        for (int32_t k = 0; k < nblocks; ++k) {
            const int32_t nl = left->size[block_il[k]], nr = right->size[block_ir[k]];
            for (int ti = 0; ti < (nl + 31) / 32; ++ti) for (int tj = 0; tj < (nr + 31) / 32; ++tj)
                tt.push_back(TrTile{off[k], psiT_off + off[k], nl, nr, ti, tj});
        }
        DevBuf d_tt;
        DMRGX_CHK(upload(d_tt, tt, st));
        // transpose reads psi (caller memory) and writes the arena: pass distinct base pointers
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)tt.size()), dim3(256), 0, st, d_tt.as<TrTile>(), psi_dev, buf);
        DMRGX_HIP(hipGetLastError());
        std::vector<GProd> prods;
        std::vector<GGroup> groups;
        std::vector<GTile> gt, gb;
        for (int32_t k = 0; k < nblocks; ++k) {
            const int32_t nl = left->size[block_il[k]], nr = right->size[block_ir[k]];
            const double* Psi = psi_dev + off[k];
            const double* PsiT = buf + psiT_off + off[k];
            const MatDesc& mL = P->mats[2 * k];
            const MatDesc& mR = P->mats[2 * k + 1];
            prods.push_back(GProd{Psi, PsiT, nr, nl, nr, GPROD_GEMM, 1.0});       // rho_L = Psi Psi^T  (:1733)
            groups.push_back(GGroup{buf + mL.a_off, mL.npad, nl, nl, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nl, nl, (nr + GG_BK - 1) / GG_BK);
            prods.push_back(GProd{PsiT, Psi, nl, nr, nl, GPROD_GEMM, 1.0});       // rho_R = Psi^T Psi  (:1734)
            groups.push_back(GGroup{buf + mR.a_off, mR.npad, nr, nr, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nr, nr, (nl + GG_BK - 1) / GG_BK);
        }
        ggemm_schedule(gt); ggemm_schedule(gb);
        DevBuf dp, dg, dt, db;
        DMRGX_CHK(upload(dp, prods, st)); DMRGX_CHK(upload(dg, groups, st)); DMRGX_CHK(upload(dt, gt, st)); DMRGX_CHK(upload(db, gb, st));
        DMRGX_CHK(ggemm_launch(db.as<GTile>(), dg.as<GGroup>(), dp.as<GProd>(), (int32_t)gb.size(), st, 1));
        DMRGX_CHK(ggemm_launch(dt.as<GTile>(), dg.as<GGroup>(), dp.as<GProd>(), (int32_t)gt.size(), st, 0));
        DMRGX_HIP(hipStreamSynchronize(st));
    }

    // ---- batched block Jacobi ------------------------------------------------------------------------------------
    std::vector<double> norms(2 * nm);
    const int rounds = std::max(1, max_nb - 1);
    int sweep = 0;
    for (; sweep < 30; ++sweep) {
        hipLaunchKernelGGL(offnorm_kernel, dim3(nm), dim3(256), 0, st, dm, buf, buf + norm_off);
        DMRGX_HIP(hipGetLastError());
        DMRGX_HIP(hipMemcpyAsync(norms.data(), buf + norm_off, norms.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        bool conv = true;
        for (int mi = 0; mi < nm; ++mi) if (norms[2 * mi] > 1e-30 * (norms[2 * mi] + norms[2 * mi + 1])) conv = false;
        if (conv) break;
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(jacobi_sub_kernel, dim3((unsigned)pairs.size()), dim3(256), 0, st, dm, d_pairs.as<PairRef>(), buf, buf + rbuf_off, r);
            hipLaunchKernelGGL(jacobi_update_kernel, dim3((unsigned)tiles.size(), 2), dim3(256), 0, st, dm, d_pairs.as<PairRef>(), d_tiles.as<TileRef>(), buf, buf + rbuf_off, r, 0);
            hipLaunchKernelGGL(jacobi_update_kernel, dim3((unsigned)tiles.size(), 1), dim3(256), 0, st, dm, d_pairs.as<PairRef>(), d_tiles.as<TileRef>(), buf, buf + rbuf_off, r, 1);
            DMRGX_HIP(hipGetLastError());
        }
    }
    P->sweeps = sweep;
    if (sweep >= 30) DMRGX_FAIL(DMRGX_ERR_NOTCONV, "rdm_create: block Jacobi did not converge in 30 sweeps");

    {
        int max_npad = JS;
        for (auto& m : P->mats) max_npad = std::max(max_npad, m.npad);
        hipLaunchKernelGGL(normalize_columns_kernel, dim3(max_npad / 64, nm), dim3(256), 0, st, dm, buf);
        DMRGX_HIP(hipGetLastError());
    }
    // ---- eigenvalues as Rayleigh quotients of the renormalised eigenvectors: lambda = |Psi^T u|^2 (rho_L) / |Psi v|^2
    //      (rho_R).  Relative accuracy ~eps instead of the c*n*eps*||rho|| of the rotated diagonal, which matters for
    //      TruncErr = 1 - sum of kept eigenvalues (include/DMRGBlockContainer.hpp:1872-1875).
    std::vector<double> rq((size_t)dtot);
    {
        std::vector<GProd> prods;
        std::vector<GGroup> groups;
        std::vector<GTile> gt, gb;
        std::vector<ColNormTask> cn;
        for (int32_t k = 0; k < nblocks; ++k) {
            const int32_t nl = left->size[block_il[k]], nr = right->size[block_ir[k]];
            const double* Psi = psi_dev + off[k];
            const double* PsiT = buf + psiT_off + off[k];
            const MatDesc& mL = P->mats[2 * k];
            const MatDesc& mR = P->mats[2 * k + 1];
            double* WL = buf + w_base + 2 * off[k];                     // n_R x n_L
            double* WR = WL + (int64_t)nl * nr;                          // n_L x n_R
            prods.push_back(GProd{PsiT, buf + mL.v_off, nl, mL.npad, nl, GPROD_GEMM, 1.0});
            groups.push_back(GGroup{WL, nl, nr, nl, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nr, nl, (nl + GG_BK - 1) / GG_BK);
            cn.push_back(ColNormTask{w_base + 2 * off[k], rq_base + (diag_off[2 * k] - diag_base), nr, nl, nl, 0});
            prods.push_back(GProd{Psi, buf + mR.v_off, nr, mR.npad, nr, GPROD_GEMM, 1.0});
            groups.push_back(GGroup{WR, nr, nl, nr, (int32_t)prods.size() - 1, (int32_t)prods.size(), 0, 0});
            ggemm_append_tiles_mixed(gb, gt, (int32_t)groups.size() - 1, nl, nr, (nr + GG_BK - 1) / GG_BK);
            cn.push_back(ColNormTask{w_base + 2 * off[k] + (int64_t)nl * nr, rq_base + (diag_off[2 * k + 1] - diag_base), nl, nr, nr, 0});
        }
        ggemm_schedule(gt); ggemm_schedule(gb);
        DevBuf dp, dg, dt, db, dc;
        DMRGX_CHK(upload(dp, prods, st)); DMRGX_CHK(upload(dg, groups, st)); DMRGX_CHK(upload(dt, gt, st)); DMRGX_CHK(upload(db, gb, st));
        DMRGX_CHK(upload(dc, cn, st));
        DMRGX_HIP(hipMemsetAsync(buf + rq_base, 0, (size_t)dtot * sizeof(double), st));
        DMRGX_CHK(ggemm_launch(db.as<GTile>(), dg.as<GGroup>(), dp.as<GProd>(), (int32_t)gb.size(), st, 1));
        DMRGX_CHK(ggemm_launch(dt.as<GTile>(), dg.as<GGroup>(), dp.as<GProd>(), (int32_t)gt.size(), st, 0));
        int maxc = 1;
        for (auto& c : cn) maxc = std::max(maxc, c.ncols);
        hipLaunchKernelGGL(colnorm_kernel, dim3((maxc + 63) / 64, (unsigned)cn.size()), dim3(256), 0, st, dc.as<ColNormTask>(), buf, buf);
        DMRGX_HIP(hipGetLastError());
        DMRGX_HIP(hipMemcpyAsync(rq.data(), buf + rq_base, rq.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
    }
    // ---- sort descending on the host, remember the column of V for each rank ------------------------------------
    P->eig.resize(nm); P->perm.resize(nm); P->perm_off.resize(nm);
    std::vector<int32_t> allperm;
    for (int mi = 0; mi < nm; ++mi) {
        const MatDesc& m = P->mats[mi];
        // Padding indices (>= n) are decoupled (zero off-diagonals, rotations with a_pq == 0 are skipped), so the real
        // eigenvectors are exactly the columns [0, n) of V and the padding columns stay unit vectors.
        const double* q = rq.data() + (diag_off[mi] - diag_base);
        std::vector<int32_t> real(m.n);
        std::iota(real.begin(), real.end(), 0);
        std::stable_sort(real.begin(), real.end(), [&](int32_t a, int32_t b) { return q[a] > q[b]; });
        P->perm[mi] = real;
        P->eig[mi].resize(m.n);
        for (int32_t r = 0; r < m.n; ++r) P->eig[mi][r] = q[real[r]];
        P->perm_off[mi] = (int64_t)allperm.size();
        allperm.insert(allperm.end(), real.begin(), real.end());
    }
    DMRGX_CHK(upload(P->d_perm, allperm, st));
    DMRGX_HIP(hipStreamSynchronize(st));
    *out = P.release();
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_eigenvalues(const dmrgx_rdm* R, int32_t side, int32_t k, double* host_out)
{
    if (!R || !host_out || side < 0 || side > 1 || k < 0 || k >= R->nblocks) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvalues: bad argument");
    const auto& e = R->eig[2 * k + side];
    std::copy(e.begin(), e.end(), host_out);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_eigenvectors(const dmrgx_rdm* R, int32_t side, int32_t k, int32_t count, double* dst_dev, int64_t ld, void* stream)
{
    if (!R || side < 0 || side > 1 || k < 0 || k >= R->nblocks || count < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: bad argument");
    const int mi = 2 * k + side;
    const MatDesc& m = R->mats[mi];
    if (count > m.n || (count > 0 && (!dst_dev || ld < m.n))) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_eigenvectors: count %d > n %d or bad destination", count, m.n);
    if (count == 0) return DMRGX_OK;
    hipLaunchKernelGGL(gather_vec_kernel, dim3((m.n + 255) / 256, count), dim3(256), 0, (hipStream_t)stream,
                       R->buf.as<double>() + m.v_off, m.npad, m.n, R->d_perm.as<int32_t>() + R->perm_off[mi], count, dst_dev, ld);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_info(const dmrgx_rdm* R, int32_t* n_sweeps)
{
    if (!R || !n_sweeps) DMRGX_FAIL(DMRGX_ERR_ARG, "rdm_info: bad argument");
    *n_sweeps = R->sweeps;
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_rdm_destroy(dmrgx_rdm* R)
{
    if (!R) return DMRGX_OK;
    (void)hipDeviceSynchronize();
    delete R;
    return DMRGX_OK;
}
