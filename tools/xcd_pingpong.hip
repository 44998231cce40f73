// Workgroup-to-workgroup exchange latency by placement: two workgroups of the SAME XCD against two of different XCDs, with the
// relaxed agent-scope 8-byte granules the persistent tridiagonalisation uses (symeig.hip), and a one-to-many variant (one
// producer, G-1 pollers, as in its per-column all-gather).  Prints the XCC_ID of every workgroup so the id -> XCD rule is seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
__device__ inline void put(u64* p, u64 v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline u64 get(const u64* p) { return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// workgroups a and b bounce a counter `iters` times; everyone else leaves at once
__global__ void pingpong(u64* cell, int a, int b, int iters, long long* ticks, int* xcc) {
    const int me = blockIdx.x;
    if (threadIdx.x == 0) xcc[me] = xcc_id();
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (me == a) { put(cell, (u64)(2 * i - 1)); int spin = 0; while (get(cell + 8) != (u64)(2 * i) && ++spin < (1 << 22)) {} }
        else         { int spin = 0; while (get(cell) != (u64)(2 * i - 1) && ++spin < (1 << 22)) {} put(cell + 8, (u64)(2 * i)); }
    }
    if (me == a) ticks[0] = wall_clock64() - t0;
}
// all-gather of one granule per workgroup among the workgroups listed in `members` (count g), `iters` rounds
__global__ void allgather(u64* cells, const int* members, int g, int iters, long long* ticks) {
    int slot = -1;
    for (int i = 0; i < g; ++i) if (members[i] == (int)blockIdx.x) slot = i;
    if (slot < 0) return;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        u64* row = cells + (size_t)(it & 1) * 64 * 8;
        if (threadIdx.x == 0) put(row + slot * 8, (u64)it);
        if ((int)threadIdx.x < g) { int spin = 0; while (get(row + threadIdx.x * 8) != (u64)it && ++spin < (1 << 22)) {} }
        __syncthreads();
    }
    if (slot == 0 && threadIdx.x == 0) ticks[0] = wall_clock64() - t0;
}
int main() {
    u64* cells; long long* ticks; int* xcc; int* members;
    hipMalloc(&cells, 64 * 8 * 2 * 8); hipMalloc(&ticks, 8); hipMalloc(&xcc, 1024 * 4); hipMalloc(&members, 64 * 4);
    const int iters = 2000, nwg = 64;
    std::vector<int> hx(nwg);
    auto run_pp = [&](int a, int b) {
        hipMemset(cells, 0, 64 * 8 * 2 * 8);
        pingpong<<<nwg, 64>>>(cells, a, b, iters, ticks, xcc);
        hipDeviceSynchronize();
        long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, nwg * 4, hipMemcpyDeviceToHost);
        printf("ping-pong workgroups %2d (XCC %d) <-> %2d (XCC %d): %.0f ns per round trip\n", a, hx[a], b, hx[b], t * 10.0 / iters);
    };
    run_pp(0, 8); run_pp(0, 16); run_pp(0, 1); run_pp(0, 4); run_pp(3, 11);
    printf("XCC of workgroups 0..23:"); for (int i = 0; i < 24; ++i) printf(" %d", hx[i]); printf("\n");
    auto run_ag = [&](std::vector<int> m, const char* what) {
        hipMemset(cells, 0, 64 * 8 * 2 * 8);
        hipMemcpy(members, m.data(), m.size() * 4, hipMemcpyHostToDevice);
        allgather<<<nwg, 64>>>(cells, members, (int)m.size(), iters, ticks);
        hipDeviceSynchronize();
        long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("all-gather of %zu workgroups, %s: %.0f ns per round\n", m.size(), what, t * 10.0 / iters);
    };
    run_ag({0, 8, 16, 24}, "one XCD");
    run_ag({0, 1, 2, 3}, "four XCDs");
    run_ag({0, 8, 16, 24, 32, 40, 48, 56}, "one XCD");
    run_ag({0, 1, 2, 3, 4, 5, 6, 7}, "eight XCDs");
    return 0;
}
