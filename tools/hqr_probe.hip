// Standalone check + timing of the batched Householder QR (csrc/hqr.hip).  Usage: hqr_probe [n] [count] [reps]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idmrg.x_amd/csrc tools/hqr_probe.hip dmrg.x_amd/csrc/hqr.hip -o tools/hqr_probe
#include "hqr.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
namespace dmrgx { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
using namespace dmrgx;
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 350, count = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<HqrMat> mats;
    int64_t total = 0;
    for (int i = 0; i < count; ++i) {
        HqrMat m{};
        m.n = std::max(1, n - 7 * i);
        m.b_off = total; total += 2LL * m.n * m.n;
        m.v_off = total; total += 32LL * m.n;
        m.t_off = total; total += 1024;
        mats.push_back(m);
    }
    std::vector<double> h((size_t)total, 0.0);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (auto& m : mats) {       // graded symmetric matrix [A | I]
        const int k = m.n;
        std::vector<double> X((size_t)k * k);
        for (auto& x : X) x = nd(rng);
        for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
            double s = 0; for (int l = 0; l < k; ++l) s += X[(size_t)i * k + l] * std::exp(-0.1 * l) * X[(size_t)j * k + l];
            h[m.b_off + (size_t)i * 2 * k + j] = s;
        }
        for (int i = 0; i < k; ++i) h[m.b_off + (size_t)i * 2 * k + k + i] = 1.0;
    }
    double* d; HqrMat* dm;
    hipMalloc(&d, total * sizeof(double)); hipMalloc(&dm, mats.size() * sizeof(HqrMat));
    hipMemcpy(dm, mats.data(), mats.size() * sizeof(HqrMat), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipMemcpy(d, h.data(), total * sizeof(double), hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        if (hqr_batched(mats, dm, d, 0)) return 1;
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    std::vector<double> out((size_t)total);
    hipMemcpy(out.data(), d, total * sizeof(double), hipMemcpyDeviceToHost);
    double worst_orth = 0, worst_tri = 0;
    for (auto& m : mats) {
        const int k = m.n;
        auto QT = [&](int r, int c) { return out[m.b_off + (size_t)r * 2 * k + k + c]; };
        for (int i = 0; i < k; i += std::max(1, k / 40)) for (int j = 0; j < k; ++j) {
            double s = 0; for (int l = 0; l < k; ++l) s += QT(i, l) * QT(j, l);
            worst_orth = std::max(worst_orth, std::fabs(s - (i == j)));
        }
        // R = Q^T A must be upper triangular: check a few rows below the diagonal
        for (int i = 1; i < k; i += std::max(1, k / 40)) for (int j = 0; j < i; j += std::max(1, i / 20)) {
            double s = 0; for (int l = 0; l < k; ++l) s += QT(i, l) * h[m.b_off + (size_t)l * 2 * k + j];
            worst_tri = std::max(worst_tri, std::fabs(s));
        }
    }
    const int panels = (n + 31) / 32;
    printf("n %d x %d matrices: %.3f ms (%.1f us per panel step), |Q^T Q - I| %.2e, below-diagonal |Q^T A| %.2e\n", n, count, best, 1e3 * best / panels, worst_orth, worst_tri);
    return (worst_orth < 1e-12 && worst_tri < 1e-12) ? 0 : 2;
}
