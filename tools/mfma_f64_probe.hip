// Probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) lane->element maps, (2) issue rate, (3) v_fma_f64 rate.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

// C(16x16) = A(16x4) * B(4x16); guide: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C col=l&15,row=(l>>4)+4*reg
__global__ void layout_k(const double* A, const double* B, double* C) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];  // raw dump: lane-major
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_k(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters / 16; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) fma_k(double* out, int iters, double seed) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  // ---- layout ----
  std::vector<double> A(64), B(64), C(256), Cref(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 131;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + k * 17 + j * 1009;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; Cref[i * 16 + j] = s; }
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_k<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  int bad_guide = 0, bad_f32map = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double v = C[l * 4 + r];
    int col = l & 15;
    int row_g = (l >> 4) + 4 * r;      // guide's f64 map
    int row_f = (l >> 4) * 4 + r;      // f32-style map
    if (v != Cref[row_g * 16 + col]) bad_guide++;
    if (v != Cref[row_f * 16 + col]) bad_f32map++;
  }
  printf("layout: mismatches guide-map=%d f32-style-map=%d (0 means that map is right)\n", bad_guide, bad_f32map);

  // ---- rates ----
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount; printf("device %s CUs=%d clock=%d kHz\n", p.name, cus, p.clockRate);
  double* dout; CK(hipMalloc(&dout, sizeof(double) * 256 * cus * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms * 1e-3; };
  int iters = 64000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu*=2) {
    int grid = cus * wgs_per_cu;
    double t1 = timeit([&] { rate_k<1><<<grid, 256>>>(dout, iters, 1.0); });
    double t4 = timeit([&] { rate_k<4><<<grid, 256>>>(dout, iters, 1.0); });
    double t8 = timeit([&] { rate_k<8><<<grid, 256>>>(dout, iters / 2, 1.0); });
    double f1 = 2048.0 * iters * 1 * 4 * grid / t1, f4 = 2048.0 * iters * 4 * 4 * grid / t4, f8 = 2048.0 * (iters / 2) * 8 * 4 * grid / t8;
    printf("mfma_f64_16x16x4 wg/cu=%d: 1acc %.2f TF  4acc %.2f TF  8acc %.2f TF\n", wgs_per_cu, f1 / 1e12, f4 / 1e12, f8 / 1e12);
  }
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    int grid = cus * wgs_per_cu;
    double t = timeit([&] { fma_k<<<grid, 256>>>(dout, iters, 1.0); });
    printf("v_fma_f64 wg/cu=%d: %.2f TF\n", wgs_per_cu, 2.0 * 16 * iters * 256.0 * grid / t / 1e12);
  }
  return 0;
}
