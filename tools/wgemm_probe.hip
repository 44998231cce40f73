// Prototype probe (not part of the library): a barrier-free f64 GEMM tile -- ONE wave per 64 x 64 output tile.
//   A (row-major [i][k]) goes global -> registers -> wave-private LDS (double-buffered, transposition only) -> MFMA fragments;
//   B (row-major [k][j]) goes global -> MFMA fragments directly (lane (k = l>>4, j = l&15) reads 4 rows x 128 B per instruction).
// No s_barrier anywhere: the two waves a SIMD holds are independent instruction streams.
// usage: wgemm_probe <K> <groups> [reps]      (every group computes C_g (128 x 128) = A (128 x K) . B (K x 128), shared operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(1)))* gptr;
typedef const d2 __attribute__((address_space(1)))* gptr2;

constexpr int BK = 16, AS_LD = 18;

// All global loads and their waits are written by hand: the loads of a step are in flight across the loop's back edge, where the
// compiler's counter analysis gives up and waits for vmcnt(0) -- i.e. for loads it issued a few instructions earlier.
// Order of the vector-memory queue in the steady state (one k-step): A(s+1) x8 | B(s+1,kg0) x4 | B(s+1,kg1) x4 | B(s+1,kg2) x4 |
// B(s+1,kg3) x4; a use of B(s,kg) therefore has 20 younger loads behind it, the park of A(s+1) into LDS after k-group 2 has 12.
#define GLOAD2(dst, voff, sbase, imm) asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(imm) : "memory")
#define GLOAD4(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ void __launch_bounds__(64, 2)
wgemm_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int M, int N, int K)
{
    __shared__ double As[2][64 * AS_LD];
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    const int tiles_n = N / 64, tiles_per = (M / 64) * tiles_n;
    const int g = blockIdx.x / tiles_per, t = blockIdx.x % tiles_per, m0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;
    d4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
    // A loader: 8 lanes x 16 B per row, 8 rows per instruction, 8 instructions per k-step
    const int a_r = lane >> 3, a_k = (lane & 7) * 2;
    unsigned aoff[8], boff[4];                       // per-lane byte offsets (loop invariants); the bases advance in SGPRs
#pragma unroll
    for (int q = 0; q < 8; ++q) aoff[q] = (unsigned)(((a_r + 8 * q) * K + a_k) * 8);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) boff[kg] = (unsigned)(((4 * kg + l4) * N + l15) * 8);
    const double* Ab = A + (size_t)m0 * K;           // wave-uniform
    const double* Bb = B + n0;
    d2 ra[8];
    double rb[4][4];
    // prologue: A(0) -> LDS buf 0, B(0) -> registers
#pragma unroll
    for (int q = 0; q < 8; ++q) GLOAD4(ra[q], aoff[q], Ab);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) { GLOAD2(rb[kg][0], boff[kg], Bb, 0); GLOAD2(rb[kg][1], boff[kg], Bb, 128); GLOAD2(rb[kg][2], boff[kg], Bb, 256); GLOAD2(rb[kg][3], boff[kg], Bb, 384); }
    VMWAIT(16);
#pragma unroll
    for (int q = 0; q < 8; ++q) { asm volatile("" : "+v"(ra[q])); *(d2*)&As[0][(a_r + 8 * q) * AS_LD + a_k] = ra[q]; }
    const int nsteps = K / BK;
    for (int s = 0; s < nsteps; ++s) {
        const int p = s & 1;
        // past the end the loaders re-read the last step and the spare LDS buffer receives data nobody reads (branch-free)
        const double* An = Ab + (size_t)min(s + 1, nsteps - 1) * BK;
        const double* Bn = Bb + (size_t)min(s + 1, nsteps - 1) * BK * N;
#pragma unroll
        for (int q = 0; q < 8; ++q) GLOAD4(ra[q], aoff[q], An);
        __builtin_amdgcn_sched_barrier(0);
        const double* as = As[p];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            double fa[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi] = as[(16 * mi + l15) * AS_LD + 4 * kg + l4];
            VMWAIT(20);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+v"(rb[kg][ni]));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], rb[kg][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // this k-group's B fragments are consumed: fetch the same k-group of the next step
            GLOAD2(rb[kg][0], boff[kg], Bn, 0); GLOAD2(rb[kg][1], boff[kg], Bn, 128); GLOAD2(rb[kg][2], boff[kg], Bn, 256); GLOAD2(rb[kg][3], boff[kg], Bn, 384);
            if (kg == 2) {                           // park A(s+1) in the other buffer (its readers, step s-1, are long done)
                VMWAIT(12);
#pragma unroll
                for (int q = 0; q < 8; ++q) { asm volatile("" : "+v"(ra[q])); *(d2*)&As[p ^ 1][(a_r + 8 * q) * AS_LD + a_k] = ra[q]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    VMWAIT(0);
    double* Cg = C + (size_t)g * M * N;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(m0 + 16 * mi + l4 + 4 * r) * N + n0 + 16 * ni + l15] = acc[mi][ni][r];
}

int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 1024, G = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 10;
    const int M = 128, N = 128;
    double *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 8)); CK(hipMalloc(&B, (size_t)K * N * 8)); CK(hipMalloc(&C, (size_t)M * N * G * 8));
    std::vector<double> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25 + 0.001 * (double)(i % 7);
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> hb((size_t)K * N);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = 0.5 - 0.002 * (double)(i % 5);
    CK(hipMemcpy(B, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)(G * (M / 64) * (N / 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wgemm_kernel, dim3(grid), dim3(64), 0, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wgemm_kernel, dim3(grid), dim3(64), 0, 0, A, B, C, M, N, K);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // check a few entries of the last group against the host
    std::vector<double> c((size_t)M * N);
    CK(hipMemcpy(c.data(), C + (size_t)(G - 1) * M * N, c.size() * 8, hipMemcpyDeviceToHost));
    double maxerr = 0.0;
    for (int i : {0, 17, 63, 64, 127}) for (int j : {0, 15, 16, 77, 127}) {
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += h[(size_t)i * K + k] * hb[(size_t)k * N + j];
        maxerr = fmax(maxerr, fabs(ref - c[(size_t)i * N + j]) / fabs(ref));
    }
    const double fl = 2.0 * M * N * (double)K * G;
    printf("wgemm K=%d groups=%d waves=%u : %.3f ms/launch  %.2f TF/s  (%.1f%% of 78.6)  max rel err %.1e\n", K, G, grid, ms / reps, fl * reps / (ms * 1e-3) / 1e12,
           fl * reps / (ms * 1e-3) / 78.6e10, maxerr);
    return 0;
}
