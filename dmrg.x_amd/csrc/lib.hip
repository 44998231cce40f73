// libdmrgx_hip.so: library-level entry points (error reporting, device probe, plain GEMM wrapper).
#include "ggemm.h"
#include <algorithm>

namespace dmrgx {
static thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace dmrgx

using namespace dmrgx;

extern "C" int32_t dmrgx_abi_version(void) { return DMRGX_ABI_VERSION; }

extern "C" const char* dmrgx_last_error(void) { return g_last_error.c_str(); }

extern "C" dmrgx_status dmrgx_device_count(int32_t* n)
{
    if (!n) DMRGX_FAIL(DMRGX_ERR_ARG, "device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *n = 0;
        DMRGX_FAIL(DMRGX_ERR_DEVICE, "no HIP device available (%s): the dmrgx hot path has no CPU fallback", hipGetErrorString(e));
    }
    *n = c;
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_stripe_bounds(int32_t n_right, int32_t world_size, int32_t rank, int32_t* c0, int32_t* c1)
{
    if (!c0 || !c1 || n_right < 0 || world_size <= 0 || rank < 0 || rank >= world_size) DMRGX_FAIL(DMRGX_ERR_ARG, "stripe_bounds: bad argument");
    *c0 = dmrgx::stripe_cut(n_right, world_size, rank);
    *c1 = dmrgx::stripe_cut(n_right, world_size, rank + 1);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_stripe_bounds_of_block(int32_t n_right, int32_t world_size, int32_t rank, int32_t block, int32_t* c0, int32_t* c1)
{
    if (!c0 || !c1 || n_right < 0 || world_size <= 0 || rank < 0 || rank >= world_size || block < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "stripe_bounds_of_block: bad argument");
    const int32_t j = dmrgx::stripe_of_rank(world_size, rank, block);
    *c0 = dmrgx::stripe_cut(n_right, world_size, j);
    *c1 = dmrgx::stripe_cut(n_right, world_size, j + 1);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_dgemm_nn(int32_t M, int32_t N, int32_t K, const double* A, int64_t lda,
                                       const double* B, int64_t ldb, double* C, int64_t ldc, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (M < 0 || N < 0 || K < 0 || (M && N && (!C || ldc < N)) || (M && N && K && (!A || !B || lda < K || ldb < N)))
        DMRGX_FAIL(DMRGX_ERR_ARG, "dgemm_nn: bad argument (M=%d N=%d K=%d)", M, N, K);
    if (M == 0 || N == 0) return DMRGX_OK;
    std::vector<GProd> prods;
    if (K > 0) prods.push_back(GProd{A, B, (int32_t)lda, (int32_t)ldb, K, GPROD_GEMM, 1.0});
    std::vector<GGroup> groups = {GGroup{C, (int32_t)ldc, M, N, 0, (int32_t)prods.size(), 0, 0}};
    std::vector<GTile> tiles, big;
    ggemm_append_tiles_mixed(big, tiles, 0, M, N, (K + GG_BK - 1) / GG_BK);
    ggemm_schedule(tiles, groups); ggemm_schedule(big, groups, 2);
    if (prods.empty()) prods.push_back(GProd{nullptr, nullptr, 0, 0, 0, GPROD_GEMM, 0.0});
    DevBuf tab;
    PackedUpload pk;
    const size_t op = pk.add(prods), og = pk.add(groups), ot = pk.add(tiles), ob = pk.add(big);
    DMRGX_CHK(pk.upload(tab, st));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(tab, ob), packed_at<GGroup>(tab, og), packed_at<GProd>(tab, op), (int32_t)big.size(), st, 1));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(tab, ot), packed_at<GGroup>(tab, og), packed_at<GProd>(tab, op), (int32_t)tiles.size(), st, 0));
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_dgemm_batch(int32_t count, const dmrgx_gemm_task* t, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (count < 0 || (count > 0 && !t)) DMRGX_FAIL(DMRGX_ERR_ARG, "dgemm_batch: bad argument");
    std::vector<GProd> prods;
    std::vector<GGroup> groups;
    std::vector<GTile> tiles, big;
    for (int32_t i = 0; i < count; ++i) {
        const dmrgx_gemm_task& g = t[i];
        if (g.M < 0 || g.N < 0 || g.K < 0 || (g.M && g.N && (!g.C || g.ldc < g.N)) || (g.M && g.N && g.K && (!g.A || !g.B || g.lda < g.K || g.ldb < g.N)))
            DMRGX_FAIL(DMRGX_ERR_ARG, "dgemm_batch: bad task %d (M=%d N=%d K=%d)", i, g.M, g.N, g.K);
        if (g.M == 0 || g.N == 0) continue;
        const int32_t p0 = (int32_t)prods.size();
        if (g.K > 0) prods.push_back(GProd{g.A, g.B, (int32_t)g.lda, (int32_t)g.ldb, g.K, GPROD_GEMM, 1.0});
        groups.push_back(GGroup{g.C, (int32_t)g.ldc, g.M, g.N, p0, (int32_t)prods.size(), 0, g.accumulate ? 1 : 0});
        ggemm_append_tiles_mixed(big, tiles, (int32_t)groups.size() - 1, g.M, g.N, (g.K + GG_BK - 1) / GG_BK);
    }
    if (groups.empty()) return DMRGX_OK;
    if (prods.empty()) prods.push_back(GProd{nullptr, nullptr, 0, 0, 0, GPROD_GEMM, 0.0});
    ggemm_schedule(tiles, groups); ggemm_schedule(big, groups, 2);
    DevBuf tab;
    PackedUpload pk;
    const size_t op = pk.add(prods), og = pk.add(groups), ot = pk.add(tiles), ob = pk.add(big);
    DMRGX_CHK(pk.upload(tab, st));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(tab, ob), packed_at<GGroup>(tab, og), packed_at<GProd>(tab, op), (int32_t)big.size(), st, 1));
    DMRGX_CHK(ggemm_launch(packed_at<GTile>(tab, ot), packed_at<GGroup>(tab, og), packed_at<GProd>(tab, op), (int32_t)tiles.size(), st, 0));
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_malloc(void** p, size_t bytes)
{
    if (!p) DMRGX_FAIL(DMRGX_ERR_ARG, "malloc: null argument");
    *p = nullptr;
    if (bytes == 0) return DMRGX_OK;
    hipError_t e = dmrgx::pool_malloc(p, bytes);
    if (e != hipSuccess) { *p = nullptr; DMRGX_FAIL(DMRGX_ERR_MEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return DMRGX_OK;
}
extern "C" dmrgx_status dmrgx_mem_stats(size_t* in_use, size_t* cached, size_t* peak_in_use)
{
    dmrgx::pool_stats(in_use, cached, peak_in_use);
    return DMRGX_OK;
}
extern "C" dmrgx_status dmrgx_free(void* p) { if (p) DMRGX_HIP(dmrgx::pool_free(p)); return DMRGX_OK; }
extern "C" dmrgx_status dmrgx_memcpy_h2d(void* d, const void* s, size_t n, void* st)
{ if (n) DMRGX_HIP(dmrgx::h2d_async(d, s, n, (hipStream_t)st)); return DMRGX_OK; }
extern "C" dmrgx_status dmrgx_memcpy_d2h(void* d, const void* s, size_t n, void* st)
{ if (n) { DMRGX_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st)); DMRGX_HIP(hipStreamSynchronize((hipStream_t)st)); } return DMRGX_OK; }
extern "C" dmrgx_status dmrgx_memcpy_d2d(void* d, const void* s, size_t n, void* st)
{ if (n) DMRGX_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st)); return DMRGX_OK; }
extern "C" dmrgx_status dmrgx_memset_zero(void* d, size_t n, void* st)
{ if (n) DMRGX_HIP(dmrgx::zero_async(d, n, (hipStream_t)st)); return DMRGX_OK; }
extern "C" dmrgx_status dmrgx_stream_sync(void* st) { DMRGX_HIP(hipStreamSynchronize((hipStream_t)st)); return DMRGX_OK; }

// ---- <x, y> on the device (correlators: VecDot of the reference, include/DMRGBlockContainer.hpp:2291) ------------------
namespace dmrgx {
namespace {
constexpr int VDOT_BLOCKS = 512, VDOT_THREADS = 256;
__global__ void __launch_bounds__(VDOT_THREADS) vdot_partial_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n, double* __restrict__ partial)
{
    __shared__ double red[VDOT_THREADS / 64];
    double s = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * VDOT_THREADS + threadIdx.x; e < n; e += (int64_t)VDOT_BLOCKS * VDOT_THREADS) s += x[e] * y[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < VDOT_THREADS / 64; ++k) t += red[k]; partial[blockIdx.x] = t; }
}
}  // namespace
}  // namespace dmrgx

namespace dmrgx {
namespace {
__global__ void __launch_bounds__(64) vdot_final_kernel(const double* __restrict__ partial, double* __restrict__ out)
{
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < VDOT_BLOCKS; ++k) t += partial[k]; out[0] = t; }    // same order as the host sum of dmrgx_dot
}
}  // namespace
}  // namespace dmrgx

extern "C" dmrgx_status dmrgx_dot_async(int64_t n, const double* x_dev, const double* y_dev, double* dev_out, void* stream)
{
    if (n < 0 || !dev_out || (n > 0 && (!x_dev || !y_dev))) DMRGX_FAIL(DMRGX_ERR_ARG, "dot_async: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { DMRGX_HIP(zero_async(dev_out, sizeof(double), st)); return DMRGX_OK; }
    DevBuf part;                                      // returned to the pool on exit; recycling is stream-ordered
    DMRGX_CHK(part.alloc(dmrgx::VDOT_BLOCKS * sizeof(double)));
    hipLaunchKernelGGL(dmrgx::vdot_partial_kernel, dim3(dmrgx::VDOT_BLOCKS), dim3(dmrgx::VDOT_THREADS), 0, st, x_dev, y_dev, n, part.as<double>());
    hipLaunchKernelGGL(dmrgx::vdot_final_kernel, dim3(1), dim3(64), 0, st, (const double*)part.as<double>(), dev_out);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_dot(int64_t n, const double* x_dev, const double* y_dev, double* host_out, void* stream)
{
    if (n < 0 || !host_out || (n > 0 && (!x_dev || !y_dev))) DMRGX_FAIL(DMRGX_ERR_ARG, "dot: bad argument");
    *host_out = 0.0;
    if (n == 0) return DMRGX_OK;
    hipStream_t st = (hipStream_t)stream;
    DevBuf part;
    DMRGX_CHK(part.alloc(dmrgx::VDOT_BLOCKS * sizeof(double)));
    hipLaunchKernelGGL(dmrgx::vdot_partial_kernel, dim3(dmrgx::VDOT_BLOCKS), dim3(dmrgx::VDOT_THREADS), 0, st, x_dev, y_dev, n, part.as<double>());
    DMRGX_HIP(hipGetLastError());
    std::vector<double> h(dmrgx::VDOT_BLOCKS);
    DMRGX_HIP(hipMemcpyAsync(h.data(), part.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    DMRGX_HIP(hipStreamSynchronize(st));
    double t = 0.0;
    for (double v : h) t += v;          // fixed order: reproducible
    *host_out = t;
    return DMRGX_OK;
}

// ---- many Frobenius inner products in one launch (correlators of operators living on one block: <psi|P (x) 1|psi> =
//      sum_k <P[IL(k)], X_k X_k^T>_F, so a table of correlators is one batch of 2-D dot products against the Gram blocks) -----
namespace dmrgx {
namespace {
struct Dot2dPiece { const double* a; const double* b; int64_t lda, ldb; int32_t nr, nc; };
__global__ void __launch_bounds__(256) dot2d_partial_kernel(const Dot2dPiece* __restrict__ pieces, double* __restrict__ partial)
{
    __shared__ double red[4];
    const Dot2dPiece t = pieces[blockIdx.x];
    double s = 0.0;
    const int64_t tot = (int64_t)t.nr * t.nc;
    for (int64_t e = threadIdx.x; e < tot; e += 256) { const int64_t i = e / t.nc, j = e % t.nc; s += t.a[i * t.lda + j] * t.b[i * t.ldb + j]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[outs[o]] = sum of partial[first[o] .. first[o+1]) in piece order
__global__ void __launch_bounds__(256) dot2d_final_kernel(const double* __restrict__ partial, const int32_t* __restrict__ outs, const int32_t* __restrict__ first, int nout, double* __restrict__ out)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    double s = 0.0;
    for (int q = first[o]; q < first[o + 1]; ++q) s += partial[q];
    out[outs[o]] = s;
}
}  // namespace
}  // namespace dmrgx

extern "C" dmrgx_status dmrgx_dot2d_batch(int32_t count, const dmrgx_dot2d_task* tasks, double* dev_out, void* stream)
{
    if (count < 0 || (count > 0 && (!tasks || !dev_out))) DMRGX_FAIL(DMRGX_ERR_ARG, "dot2d_batch: bad argument");
    if (count == 0) return DMRGX_OK;
    hipStream_t st = (hipStream_t)stream;
    // pieces of at most ~64k elements, grouped by output index (stable: task order inside an output is kept)
    std::vector<int32_t> order(count);
    for (int32_t i = 0; i < count; ++i) {
        const dmrgx_dot2d_task& t = tasks[i];
        if (t.nr < 0 || t.nc < 0 || t.out < 0 || (t.nr && t.nc && (!t.a || !t.b || t.lda < t.nc || t.ldb < t.nc))) DMRGX_FAIL(DMRGX_ERR_ARG, "dot2d_batch: bad task %d", i);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return tasks[x].out < tasks[y].out; });
    std::vector<dmrgx::Dot2dPiece> pieces;
    std::vector<int32_t> outs, first;
    for (int32_t r = 0; r < count; ++r) {
        const dmrgx_dot2d_task& t = tasks[order[r]];
        if (outs.empty() || outs.back() != t.out) { outs.push_back(t.out); first.push_back((int32_t)pieces.size()); }
        if (t.nr == 0 || t.nc == 0) continue;
        const int32_t rows_per = std::max<int32_t>(1, 65536 / t.nc);
        for (int32_t r0 = 0; r0 < t.nr; r0 += rows_per)
            pieces.push_back(dmrgx::Dot2dPiece{t.a + (int64_t)r0 * t.lda, t.b + (int64_t)r0 * t.ldb, t.lda, t.ldb, std::min(rows_per, t.nr - r0), t.nc});
    }
    first.push_back((int32_t)pieces.size());
    DevBuf tab, dpart;
    DMRGX_CHK(dpart.alloc(std::max<size_t>(pieces.size(), 1) * sizeof(double)));
    dmrgx::PackedUpload pk;
    const size_t opc = pk.add(pieces), oo = pk.add(outs), of = pk.add(first);
    DMRGX_CHK(pk.upload(tab, st));
    if (!pieces.empty()) hipLaunchKernelGGL(dmrgx::dot2d_partial_kernel, dim3((unsigned)pieces.size()), dim3(256), 0, st, dmrgx::packed_at<dmrgx::Dot2dPiece>(tab, opc), dpart.as<double>());
    hipLaunchKernelGGL(dmrgx::dot2d_final_kernel, dim3((unsigned)((outs.size() + 255) / 256)), dim3(256), 0, st, (const double*)dpart.as<double>(), (const int32_t*)dmrgx::packed_at<int32_t>(tab, oo), (const int32_t*)dmrgx::packed_at<int32_t>(tab, of),
                       (int)outs.size(), dev_out);
    DMRGX_HIP(hipGetLastError());
    return DMRGX_OK;
}
