// Clock / barrier / LDS-latency probe under light load (one or few workgroups): how long latency-bound helper kernels really take.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nops(int iters) { for (int i = 0; i < iters; ++i) { asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15" ::: ); } }
__global__ void barriers(int iters) { for (int i = 0; i < iters; ++i) __syncthreads(); }
__global__ void ldschain(int iters, int* out) {
    __shared__ int ring[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ring[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x & 1023;
    for (int i = 0; i < iters; ++i) p = ring[p];
    if (p == 12345) out[0] = p;
}
__global__ void fmachain(int iters, double* out) {
    double a = threadIdx.x * 1e-9, b = 1.0000001;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
    if (a == 0.12345) out[0] = a;
}
__global__ void bperm(int iters, double* out) {
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a += __shfl_xor(a, 8, 64);
    if (a == 0.12345) out[0] = a;
}
template <class F> float timeit(F f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }
int main() {
    int* di; double* dd; hipMalloc(&di, 64); hipMalloc(&dd, 64);
    const int N = 100000;
    for (int wgs : {1, 16, 1024}) {
        float t1 = timeit([&] { nops<<<wgs, 64>>>(N); });
        printf("%4d WGs: 64 cycles of s_nop: %.1f ns  -> %.2f GHz\n", wgs, t1 * 1e6 / N, 64.0 * N / (t1 * 1e6));
        for (int th : {256, 512, 1024}) { float t = timeit([&] { barriers<<<wgs, th>>>(N); }); printf("         __syncthreads (%4d threads): %.1f ns\n", th, t * 1e6 / N); }
        float t3 = timeit([&] { ldschain<<<wgs, 64>>>(N, di); }); printf("         dependent LDS read: %.1f ns\n", t3 * 1e6 / N);
        float t4 = timeit([&] { fmachain<<<wgs, 64>>>(N, dd); }); printf("         dependent f64 FMA: %.1f ns\n", t4 * 1e6 / N);
        float t5 = timeit([&] { bperm<<<wgs, 64>>>(N, dd); }); printf("         dependent shfl_xor(f64)+add: %.1f ns\n", t5 * 1e6 / N);
    }
    return 0;
}
