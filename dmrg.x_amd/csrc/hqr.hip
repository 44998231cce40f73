// Batched Householder QR with explicit Q^T -- see hqr.h.  Two kernels per panel of 32 columns:
//
//   hqr_panel_*_kernel   one workgroup per matrix factors the n_rem x 32 panel (32 reflectors, one after the other) and leaves
//                        the reflectors V (explicit unit lower trapezoid) and the compact-WY factor T:
//                        H_0 .. H_31 = I - V T V^T.  Panels of up to 1024 rows live in registers (hqr_panel_regs_kernel),
//                        longer ones in the L2-resident V scratch (hqr_panel_kernel).
//   hqr_trailing_kernel  C <- (I - V T^T V^T) C for the columns right of the panel and for the n columns of the
//                        identity half; one workgroup per 64 columns, two passes over the column strip (W = V^T C reduced
//                        over eight waves, then C -= V (T^T W)); V rows are wave-uniform operands of the FMAs.
//
// Q^T = H_k .. H_1 I is accumulated forwards in the identity half, next to the factorisation (every panel updates all n of
// its columns, rows r0 and below): 2 n^3 flops on top of the 4/3 n^3 of the factorisation, no second sweep over the panels.
#include "hqr.h"
#include <type_traits>

namespace dmrgx {
namespace {

constexpr int HR_THREADS = 512, HR_WAVES = HR_THREADS / 64, HR_MAX_ROWS = 1024;     // register-resident panels
constexpr int HQ_THREADS = 1024, HQ_WAVES = HQ_THREADS / 64, HQ_LDS_ROWS = 480, HQ_LDS_STRIDE = 33;
static_assert(HQR_MAX_N <= HQ_LDS_ROWS * HQ_LDS_STRIDE, "the reflector of the global-memory path is kept in the panel's LDS");

__global__ void __launch_bounds__(HQ_THREADS)
hqr_panel_kernel(const HqrMat* __restrict__ mats, double* __restrict__ buf, int r0)
{
    __shared__ double pl[HQ_LDS_ROWS * HQ_LDS_STRIDE];     // the panel (LDS path) or the current reflector (global path)
    __shared__ double vsm[HQ_LDS_ROWS];                    // the current reflector (LDS path)
    __shared__ double red[32][33];
    __shared__ double Tm[32][33];
    __shared__ double wz[32], wred[HQ_WAVES];
    const HqrMat m = mats[blockIdx.x];
    if (m.n - r0 <= HR_MAX_ROWS) return;                    // those panels belong to hqr_panel_regs_kernel
    const int n = m.n, nrem = n - r0, pw = min(32, nrem), ldb = 2 * n;
    const int tid = threadIdx.x, c = tid & 31, rg = tid >> 5, lane = tid & 63, wave = tid >> 6;
    const double* B = buf + m.b_off + (int64_t)r0 * ldb + r0;
    double* Vg = buf + m.v_off;
    const bool in_lds = nrem <= HQ_LDS_ROWS;
    double* P = in_lds ? pl : Vg;
    const int ps = in_lds ? HQ_LDS_STRIDE : 32;
    double* v = in_lds ? vsm : pl;

    for (int i = rg; i < nrem; i += 32) P[i * ps + c] = (c < pw) ? B[(int64_t)i * ldb + c] : 0.0;
    for (int e = tid; e < 32 * 33; e += HQ_THREADS) (&Tm[0][0])[e] = 0.0;
    __syncthreads();

    for (int j = 0; j < pw; ++j) {
        // ---- the reflector of column j:  H = I - tau v v^T,  v_j = 1,  H x = beta e_j
        double s = 0.0;
        for (int i = j + 1 + tid; i < nrem; i += HQ_THREADS) { const double x = P[i * ps + j]; s += x * x; }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) wred[wave] = s;
        __syncthreads();
        double sigma = 0.0;
        for (int w = 0; w < HQ_WAVES; ++w) sigma += wred[w];
        const double alpha = P[j * ps + j];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            const double nrm = sqrt(alpha * alpha + sigma);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        __syncthreads();                                   // everyone holds alpha and sigma before column j is rewritten
        for (int i = j + tid; i < nrem; i += HQ_THREADS) {
            if (i == j) { v[i] = 1.0; P[j * ps + j] = beta; }
            else { const double vi = P[i * ps + j] * scale; v[i] = vi; P[i * ps + j] = vi; }
        }
        __syncthreads();
        // ---- one pass gives w_c = v^T P[:,c] for the columns right of j and z_c = V[:,c]^T v for the reflectors left of it
        double acc = 0.0;
        if (c != j) for (int i = j + rg; i < nrem; i += 32) acc += v[i] * P[i * ps + c];
        red[rg][c] = acc;
        __syncthreads();
        if (tid < 32) { double t = 0.0; for (int g = 0; g < 32; ++g) t += red[g][tid]; wz[tid] = t; }
        __syncthreads();
        if (c > j && c < pw) {
            const double tw = tau * wz[c];
            for (int i = j + rg; i < nrem; i += 32) P[i * ps + c] -= tw * v[i];
        }
        if (tid < 32) Tm[tid][j] = tid < j ? wz[tid] : (tid == j ? tau : 0.0);      // column j of S: z above the diagonal, tau on it
        __syncthreads();
    }
    // ---- explicit V (unit diagonal, zeros above it and right of the panel) and T for the trailing update
    for (int i = rg; i < nrem; i += 32) {
        double x = 0.0;
        if (c < pw) x = (i > c) ? P[i * ps + c] : (i == c ? 1.0 : 0.0);
        Vg[(int64_t)i * 32 + c] = x;
    }
    for (int e = tid; e < 32 * 32; e += HQ_THREADS) buf[m.t_off + e] = Tm[e >> 5][e & 31];
}

// Register-resident panel factorisation (n_rem <= 64 RR): 512 threads as 64 row blocks x 8 column blocks, thread (rb, cb)
// keeps the rows rb, rb+64, .. of the four panel columns 4cb..4cb+3 in registers, so one reflector element read from LDS
// feeds eight FMAs (the first version, one column per thread, was bound by LDS reads of the reflector).  Three barriers
// per column: the norm of the next column is accumulated while the current reflector is applied, and column j of T is
// finished one iteration late.  One body per size class (rows per thread), selected per matrix inside one launch.
// a(lane) + a(lane ^ 8) + .. over lane bits 3, 4, 5 without the LDS crossbar: DPP row rotate, then the gfx950 row / half swaps
// (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap: upper half <-> lower half;
// with both operands equal, result[0] + result[1] = own + partner in every lane).
__device__ __forceinline__ double sum_lane_bits_345(double a)
{
    int lo = __double2loint(a), hi = __double2hiint(a);
    a += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false));   // row_ror:8
    lo = __double2loint(a); hi = __double2hiint(a);
    auto l16 = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
    a = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
    lo = __double2loint(a); hi = __double2hiint(a);
    auto l32 = __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
    return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}

__device__ __forceinline__ double sel4(const double (&x)[4], int q) { return q == 0 ? x[0] : (q == 1 ? x[1] : (q == 2 ? x[2] : x[3])); }

template <int RR>
__device__ __forceinline__ void hqr_panel_regs_body(const HqrMat& m, double* __restrict__ buf, int r0, double* vsm, double (*red)[32], double (*Tm)[33], double* sred, double* piv)
{
    const int nrem = m.n - r0;
    const int pw = min(32, nrem), ldb = 2 * m.n;
    const int tid = threadIdx.x, cb = tid & 7, rb = tid >> 3, lane = tid & 63, wave = tid >> 6;
    const double* B = buf + m.b_off + (int64_t)r0 * ldb + r0 + 4 * cb;
    double p[RR][4];
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        const int i = rb + 64 * k;
#pragma unroll
        for (int q = 0; q < 4; ++q) p[k][q] = (i < nrem && 4 * cb + q < pw) ? B[(int64_t)i * ldb + q] : 0.0;
    }
    for (int e = tid; e < 32 * 33; e += HR_THREADS) (&Tm[0][0])[e] = 0.0;
    {   // norm and pivot of column 0
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < RR; ++k) { const int i = rb + 64 * k; if (i > 0) s += p[k][0] * p[k][0]; }
        s = sum_lane_bits_345(s);
        if (lane == 0) sred[wave] = s;
        if (tid == 0) piv[0] = p[0][0];
    }
    __syncthreads();
    for (int j = 0; j < pw; ++j) {
        double sigma = 0.0;
#pragma unroll
        for (int w = 0; w < HR_WAVES; ++w) sigma += sred[w];
        const double alpha = piv[0];
        double tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            const double nrm = sqrt(alpha * alpha + sigma);
            const double beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        const int k0 = j >> 6, cbj = j >> 2, qj = j & 3;     // registers k < k0 hold rows above j
        if (cb == cbj) {                                    // the reflector: zeros above row j, 1 on it, x * scale below
            // qj is uniform over the workgroup: four copies with a compile-time column index instead of select chains
            auto write_v = [&](auto Q) {
                constexpr int q = decltype(Q)::value;
#pragma unroll
                for (int k = 0; k < RR; ++k) {
                    if (k < k0) continue;
                    const int i = rb + 64 * k;
                    const double vi = (i < j) ? 0.0 : (i == j ? 1.0 : p[k][q] * scale);
                    if (i > j) p[k][q] = vi;
                    vsm[i] = vi;
                }
            };
            switch (qj) {
                case 0: write_v(std::integral_constant<int, 0>{}); break;
                case 1: write_v(std::integral_constant<int, 1>{}); break;
                case 2: write_v(std::integral_constant<int, 2>{}); break;
                default: write_v(std::integral_constant<int, 3>{}); break;
            }
        }
        __syncthreads();
        // ---- one pass gives w_c = v^T P[:,c] for the columns right of j and z_c = V[:,c]^T v for the reflectors left of it
        double vr[RR], acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < RR; ++k) {
            vr[k] = (k >= k0) ? vsm[rb + 64 * k] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += vr[k] * p[k][q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = sum_lane_bits_345(acc[q]);
            if (lane < 8) red[wave][4 * cb + q] = acc[q];
        }
        __syncthreads();
        double tw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double wc = 0.0;
#pragma unroll
            for (int w = 0; w < HR_WAVES; ++w) wc += red[w][4 * cb + q];
            const int col = 4 * cb + q;
            if (tid < 8) Tm[col][j] = col < j ? wc : (col == j ? tau : 0.0);      // column j of S: z above the diagonal, tau on it
            tw[q] = (col > j && col < pw) ? tau * wc : 0.0;
        }
#pragma unroll
        for (int k = 0; k < RR; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) p[k][q] -= tw[q] * vr[k];
        }
        if (j + 1 < pw) {                                   // norm and pivot of the next column, while it is in hand
            const int q1 = (j + 1) & 3, cb1 = (j + 1) >> 2;
            double s = 0.0;
            auto norm_next = [&](auto Q) {
                constexpr int q = decltype(Q)::value;
#pragma unroll
                for (int k = 0; k < RR; ++k) {
                    const int i = rb + 64 * k;
                    const double x = p[k][q];
                    if (i > j + 1) s += x * x; else if (i == j + 1 && cb == cb1) piv[0] = x;
                }
            };
            switch (q1) {
                case 0: norm_next(std::integral_constant<int, 0>{}); break;
                case 1: norm_next(std::integral_constant<int, 1>{}); break;
                case 2: norm_next(std::integral_constant<int, 2>{}); break;
                default: norm_next(std::integral_constant<int, 3>{}); break;
            }
            s = sum_lane_bits_345(s);
            if (lane == cb1) sred[wave] = s;
        }
        __syncthreads();
    }
    double* Vg = buf + m.v_off;
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        const int i = rb + 64 * k;
        if (i < nrem) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int c = 4 * cb + q; Vg[(int64_t)i * 32 + c] = (c < pw) ? (i > c ? p[k][q] : (i == c ? 1.0 : 0.0)) : 0.0; }
        }
    }
    for (int e = tid; e < 32 * 32; e += HR_THREADS) buf[m.t_off + e] = Tm[e >> 5][e & 31];
}

// One launch for all size classes: kernels of one stream run one after the other, so a launch per class made a panel step cost
// the SUM of the classes' latencies (matrices of several sizes are in flight in almost every step).
__global__ void __launch_bounds__(HR_THREADS)
hqr_panel_regs_kernel(const HqrMat* __restrict__ mats, double* __restrict__ buf, int r0)
{
    __shared__ double vsm[HR_MAX_ROWS];
    __shared__ double red[HR_WAVES][32];
    __shared__ double Tm[32][33];
    __shared__ double sred[HR_WAVES], piv[1];
    const HqrMat m = mats[blockIdx.x];
    const int nrem = __builtin_amdgcn_readfirstlane(m.n - r0);
    if (nrem <= 0 || nrem > HR_MAX_ROWS) return;          // nothing left, or a panel of hqr_panel_kernel
    if (nrem <= 256) hqr_panel_regs_body<4>(m, buf, r0, vsm, red, Tm, sred, piv);
    else if (nrem <= 512) hqr_panel_regs_body<8>(m, buf, r0, vsm, red, Tm, sred, piv);
    else hqr_panel_regs_body<16>(m, buf, r0, vsm, red, Tm, sred, piv);
}



// The same update on the MFMA pipe (v_mfma_f64_16x16x4): one workgroup per TM_COLS columns, eight waves.
//   pass 1: W0 (32 x TM_COLS) = V^T C -- the waves take the rows four at a time (one MFMA k-step: A = V^T fragment
//           [reflector l15][row l4], B = C fragment [row l4][column l15], both straight from global memory), 2 x TM_NBW accumulator
//           blocks per wave, summed over the waves through LDS;
//   solve : W = S'^-T W0 column by column (one thread per column);
//   pass 2: C -= V W -- the waves take 16-row blocks: A = V fragment [row l15][reflector l4] from global memory,
//           B = W from LDS, TM_NBW column blocks of eight k-steps each.
typedef double hd4 __attribute__((ext_vector_type(4)));
constexpr int TM_THREADS = 512, TM_WAVES = TM_THREADS / 64, TM_WLD = 64 + 4;
constexpr int TM_NBW = 2, TM_COLS = 16 * TM_NBW;      // strip width: 32 columns (64 measured 55 us per launch at n = 1150: one workgroup per strip sweeps all
                                                     // rows twice, so narrower strips = shorter chains on twice as many CUs)

__global__ void __launch_bounds__(TM_THREADS)
hqr_trailing_mfma_kernel(const HqrMat* __restrict__ mats, double* __restrict__ buf, const double* __restrict__ vt, int r0)
{
    __shared__ double racc[TM_WAVES][TM_NBW][256];     // one m-block row of partial W0 per wave
    __shared__ double Ws[32 * TM_WLD];
    __shared__ double Ts[32 * 32];
    const HqrMat m = mats[blockIdx.y];
    if (r0 >= m.n) return;
    const int n = m.n, nrem = n - r0, pw = min(32, nrem), ldb = 2 * n;
    const int nA = (n - r0 - pw + TM_COLS - 1) / TM_COLS, nI = (n + TM_COLS - 1) / TM_COLS;
    const int t = blockIdx.x;
    if (t >= nA + nI) return;
    int col0, cend;
    if (t < nA) { col0 = r0 + pw + t * TM_COLS; cend = n; }
    else { col0 = n + (t - nA) * TM_COLS; cend = 2 * n; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double* C = buf + m.b_off + (int64_t)r0 * ldb;
    const double* V = vt + m.v_off;
    for (int e = tid; e < 32 * 32; e += TM_THREADS) Ts[e] = vt[m.t_off + e];
    // columns of this lane in the column blocks (clamped: invalid columns are computed on garbage and never stored)
    int colb[TM_NBW]; bool cok[TM_NBW];
#pragma unroll
    for (int nb = 0; nb < TM_NBW; ++nb) { const int c = col0 + 16 * nb + l15; cok[nb] = c < cend; colb[nb] = cok[nb] ? c : col0; }

    // ---- pass 1
    hd4 acc[2][TM_NBW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < TM_NBW; ++nb) acc[mb][nb] = (hd4){0.0, 0.0, 0.0, 0.0};
    const int ksteps = (nrem + 3) / 4;
    for (int ks = wave; ks < ksteps; ks += TM_WAVES) {
        const int r = 4 * ks + l4;
        const bool rok = r < nrem;
        const int rr = rok ? r : 0;
        double a[2], b[TM_NBW];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) { const double x = V[(int64_t)rr * 32 + 16 * mb + l15]; a[mb] = rok ? x : 0.0; }
#pragma unroll
        for (int nb = 0; nb < TM_NBW; ++nb) { const double x = C[(int64_t)rr * ldb + colb[nb]]; b[nb] = rok ? x : 0.0; }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < TM_NBW; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
    }
    // reduce over the waves, one m-block at a time: W0[16 mb + l4 + 4 r][16 nb + l15]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < TM_NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[wave][nb][64 * r + lane] = acc[mb][nb][r];
        __syncthreads();
        for (int e = tid; e < TM_NBW * 256; e += TM_THREADS) {
            const int nb = e >> 8, q = e & 255, r = q >> 6, ln = q & 63;
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < TM_WAVES; ++w) sum += racc[w][nb][q];
            Ws[(16 * mb + (ln >> 4) + 4 * r) * TM_WLD + 16 * nb + (ln & 15)] = sum;
        }
    }
    __syncthreads();
    // ---- W = T^T W0 as a forward substitution with S (see hqr_trailing_kernel), one thread per column
    if (tid < TM_COLS) {
        double w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) w[k] = Ws[k * TM_WLD + tid];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            w[k] *= Ts[k * 32 + k];
#pragma unroll
            for (int l = k + 1; l < 32; ++l) w[l] -= Ts[k * 32 + l] * w[k];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) Ws[k * TM_WLD + tid] = w[k];
    }
    __syncthreads();
    // ---- pass 2
    const int rblocks = (nrem + 15) / 16;
    for (int rb = wave; rb < rblocks; rb += TM_WAVES) {
        const int ra = 16 * rb + l15;                 // A-fragment row of this lane
        const bool aok = ra < nrem;
        double a[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { const double x = V[(int64_t)(aok ? ra : 0) * 32 + 4 * g + l4]; a[g] = aok ? x : 0.0; }
#pragma unroll
        for (int nb = 0; nb < TM_NBW; ++nb) {
            hd4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < 8; ++g) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], Ws[(4 * g + l4) * TM_WLD + 16 * nb + l15], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + l4 + 4 * r;
                if (row < nrem && cok[nb]) C[(int64_t)row * ldb + colb[nb]] -= d[r];
            }
        }
    }
}

}  // namespace

dmrgx_status hqr_batched(const std::vector<HqrMat>& mats, const HqrMat* d_mats, double* buf, hipStream_t st)
{
    int max_n = 0;
    for (const HqrMat& m : mats) {
        if (m.n < 0 || m.n > HQR_MAX_N) DMRGX_FAIL(DMRGX_ERR_ARG, "hqr: matrix of order %d (supported: 0..%d)", m.n, HQR_MAX_N);
        max_n = std::max(max_n, m.n);
    }
    const unsigned nm = (unsigned)mats.size();
    if (nm == 0) return DMRGX_OK;
    for (int r0 = 0; r0 < max_n; r0 += 32) {
        const int nrem = max_n - r0, pw = std::min(32, nrem);
        const int strip = TM_COLS;
        const unsigned tiles = (unsigned)((max_n - r0 - pw + strip - 1) / strip + (max_n + strip - 1) / strip);
        bool regs = false, longp = false;                    // panels of up to HR_MAX_ROWS rows (registers) / longer ones
        for (const HqrMat& m : mats) { const int r = m.n - r0; if (r > HR_MAX_ROWS) longp = true; else if (r > 0) regs = true; }
        if (longp) hipLaunchKernelGGL(hqr_panel_kernel, dim3(nm), dim3(HQ_THREADS), 0, st, d_mats, buf, r0);
        if (regs) hipLaunchKernelGGL(hqr_panel_regs_kernel, dim3(nm), dim3(HR_THREADS), 0, st, d_mats, buf, r0);
        hipLaunchKernelGGL(hqr_trailing_mfma_kernel, dim3(tiles, nm), dim3(TM_THREADS), 0, st, d_mats, buf, (const double*)buf, r0);
        DMRGX_HIP(hipGetLastError());
    }
    return DMRGX_OK;
}

}  // namespace dmrgx
