// Pointer-chase latency probe (developer tool): dependent 8-byte global loads over a ring of `n` lines with stride
// `stride` bytes; reports ns per load for footprints that sit in L1 / L2 / MALL / HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void chase(const unsigned long long* __restrict__ ring, int iters, unsigned long long* out)
{
    unsigned long long i = threadIdx.x + blockIdx.x;   // all lanes chase the same chain (lane 0 matters)
    i = 0;
    for (int k = 0; k < iters; ++k) i = __builtin_nontemporal_load(ring + i);
    out[0] = i;
}
int main()
{
    const size_t sizes[] = {16u << 10, 256u << 10, 2u << 20, 16u << 20, 128u << 20, 1024u << 20};
    for (size_t bytes : sizes) {
        const size_t stride = 256, n = bytes / stride;
        std::vector<unsigned long long> h(bytes / 8, 0);
        // random single-cycle permutation over the n slots
        std::vector<size_t> perm(n);
        for (size_t i = 0; i < n; ++i) perm[i] = i;
        srand(1);
        for (size_t i = n - 1; i > 0; --i) { size_t j = rand() % i; std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < n; ++i) h[perm[i] * (stride / 8)] = perm[(i + 1) % n] * (stride / 8);
        unsigned long long *d, *o;
        CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 8));
        CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
        const int iters = 20000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        chase<<<1, 64>>>(d, iters, o); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); chase<<<1, 64>>>(d, iters, o); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("footprint %8zu KiB : %.1f ns per dependent load\n", bytes >> 10, ms * 1e6 / iters);
        CK(hipFree(d)); CK(hipFree(o));
    }
    return 0;
}
