// Steady-state probe for the grouped GEMM kernel (test/bench tool; links the library's ggemm.o).
//   mode L2  : every group reads the same small A (128 x K) and B (K x 128): operands stay in L2 -> pipeline efficiency
//   mode HBM : every group has its own operands -> streaming behaviour
// usage: ggemm_probe <K> <groups> <big 0|1> <shared 0|1> [reps] [products per group, each of depth K]
#include "../dmrg.x_amd/csrc/ggemm.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dmrgx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 1024, G = argc > 2 ? atoi(argv[2]) : 1024, big = argc > 3 ? atoi(argv[3]) : 0;
    const int shared = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 10, nprod = argc > 6 ? atoi(argv[6]) : 1;
    const int M = getenv("PROBE_M") ? atoi(getenv("PROBE_M")) : 128, N = getenv("PROBE_N") ? atoi(getenv("PROBE_N")) : 128;
    const size_t na = (size_t)M * K, nb = (size_t)K * N, nc = (size_t)M * N;
    const size_t copies = (shared ? 1 : G) * (size_t)nprod;
    double *A, *B, *C;
    CK(hipMalloc(&A, na * copies * 8)); CK(hipMalloc(&B, nb * copies * 8)); CK(hipMalloc(&C, nc * G * 8));
    std::vector<double> h(na * copies, 0.5);
    CK(hipMemcpy(A, h.data(), na * copies * 8, hipMemcpyHostToDevice));
    h.assign(nb * copies, 0.25);
    CK(hipMemcpy(B, h.data(), nb * copies * 8, hipMemcpyHostToDevice));
    std::vector<GProd> prods; std::vector<GGroup> groups; std::vector<GTile> tiles, tb;
    for (int g = 0; g < G; ++g) {
        for (int q = 0; q < nprod; ++q) {
            const size_t o = (shared ? 0 : (size_t)g * nprod) + q;
            prods.push_back(GProd{A + o * na, B + o * nb, K, N, K, GPROD_GEMM, 1.0});
        }
        groups.push_back(GGroup{C + (size_t)g * nc, N, M, N, g * nprod, (g + 1) * nprod, 0, 0});
        if (big) ggemm_append_tiles_mixed(tb, tiles, g, M, N, nprod * K / 16); else ggemm_append_tiles(tiles, g, M, N, nprod * K / 16);
    }
    std::vector<GTile>& tl = big ? tb : tiles;
    ggemm_schedule(tl, groups, big ? 2 : 1);
    GProd* dp; GGroup* dg; GTile* dt;
    CK(hipMalloc(&dp, prods.size() * sizeof(GProd))); CK(hipMalloc(&dg, groups.size() * sizeof(GGroup))); CK(hipMalloc(&dt, tl.size() * sizeof(GTile)));
    CK(hipMemcpy(dp, prods.data(), prods.size() * sizeof(GProd), hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, groups.data(), groups.size() * sizeof(GGroup), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tl.data(), tl.size() * sizeof(GTile), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) ggemm_launch(dt, dg, dp, (int)tl.size(), 0, big);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) ggemm_launch(dt, dg, dp, (int)tl.size(), 0, big);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * M * N * (double)K * G * nprod;
    std::vector<double> c(4); CK(hipMemcpy(c.data(), C, 32, hipMemcpyDeviceToHost));
    printf("M=%d N=%d K=%d x%d groups=%d tiles=%zu big=%d shared=%d : %.3f ms/launch  %.2f TF/s  (%.1f%% of 78.6)  c00=%g (expect %g)\n", M, N, K, nprod, G, tl.size(), big, shared, ms / reps, fl * reps / (ms * 1e-3) / 1e12, fl * reps / (ms * 1e-3) / 78.6e10, c[0], 0.125 * K * nprod);
    return 0;
}
