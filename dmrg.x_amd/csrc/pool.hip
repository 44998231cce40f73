// Device-memory pool and fill kernel of libdmrgx_hip.so.
//
// A sweep step creates and drops thousands of small device buffers (operator cells, task tables, work arrays).  hipFree
// synchronises the device and hipMemsetAsync costs ~80 us of host time per call on this stack; together they kept the GPU
// idle for half of a small-m step.  Freed blocks are therefore cached by size class and handed out again without touching
// the driver.  Recycling is stream-ordered: every operation of the library that touches a block is enqueued on a stream, so
// a block freed while work is still queued is only overwritten by work queued later ON THE SAME STREAM.  The engine and the
// Python wrappers use one stream (the null stream); a caller that moves between streams synchronises in between
// (dmrgx_stream_sync) -- see include/dmrgx.h.  DMRGX_POOL=0 disables caching (every free is a hipFree again).
#include "common.h"
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace dmrgx {
namespace {

struct Pool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_by_size;
    std::unordered_map<void*, size_t> size_of;       // every block handed out or cached
    std::unordered_map<void*, uint64_t> freed_at;    // cached blocks: value of `clock` when they came back (LRU order of the trim)
    uint64_t clock = 0;
    size_t cached = 0;
    size_t in_use = 0, peak_in_use = 0;              // bytes handed out (by size class) now / at most so far
    const bool enabled = !(getenv("DMRGX_POOL") && atoi(getenv("DMRGX_POOL")) == 0);
    const size_t cache_limit = (size_t)(getenv("DMRGX_POOL_LIMIT_GB") ? atof(getenv("DMRGX_POOL_LIMIT_GB")) : 64.0) << 30;

    static size_t size_class(size_t n) {
        if (n <= 512) return 512;
        if (n <= ((size_t)1 << 20)) { size_t c = 512; while (c < n) c <<= 1; return c; }          // powers of two up to 1 MiB
        // above: eight classes per octave (12.5 % steps), so that arenas whose size drifts a little from one sweep step to the
        // next fall into the same class and are interchangeable
        size_t oct = (size_t)1 << 20;
        while ((oct << 1) < n) oct <<= 1;                   // oct < n <= 2 oct
        const size_t step = oct / 8;
        return oct + (n - oct + step - 1) / step * step;
    }
    // Give cached blocks back to the driver, least recently freed first, until at most `target` bytes stay cached (0 =
    // everything: the out-of-memory path).  Incremental on purpose: the blocks a sweep step recycles every few milliseconds
    // are the most recently freed ones and stay; what goes are the sizes the run has moved away from.  hipFree waits for
    // the device by itself, so there is no explicit device-wide synchronisation here.
    void trim_locked(size_t target) {
        std::vector<std::pair<uint64_t, void*>> order;
        for (auto& kv : free_by_size) for (void* p : kv.second) order.push_back({freed_at[p], p});
        std::sort(order.begin(), order.end());
        for (auto& o : order) {
            if (cached <= target) break;
            void* p = o.second;
            const size_t sz = size_of[p];
            auto& v = free_by_size[sz];
            v.erase(std::find(v.begin(), v.end(), p));
            if (v.empty()) free_by_size.erase(sz);
            size_of.erase(p); freed_at.erase(p);
            cached -= sz;
            (void)hipFree(p);
        }
    }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }    // leaked on purpose: no teardown order issues with the HIP runtime

__global__ void __launch_bounds__(256) zero_kernel(uint64_t* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

}  // namespace

hipError_t pool_malloc(void** out, size_t bytes)
{
    Pool& P = pool();
    *out = nullptr;
    if (bytes == 0) return hipSuccess;
    if (!P.enabled) return hipMalloc(out, bytes);
    const size_t c = Pool::size_class(bytes);
    std::lock_guard<std::mutex> lock(P.mu);
    // best fit: the smallest cached block that holds the request; large requests may take a block up to 50 % larger (the
    // arenas of a sweep change size a little from step to step, and a driver allocation of a GiB costs ~15 ms)
    // (round 3: 25 % -> 50 %.  With 25 % an m = 2048 sweep still went to the driver 45 times for 112-128 MiB blocks, 3.9 ms each, with
    //  40 GiB of other classes cached -- one of them sat in front of the persistent tridiagonalisation of a traced step as 3.6 ms of idle GPU.)
    const size_t slack = c >= ((size_t)1 << 20) ? c / 2 : 0;
    for (auto it = P.free_by_size.lower_bound(c); it != P.free_by_size.end() && it->first <= c + slack; ++it) {
        if (it->second.empty()) continue;
        *out = it->second.back();
        it->second.pop_back();
        P.cached -= it->first;
        P.freed_at.erase(*out);
        P.in_use += it->first; P.peak_in_use = std::max(P.peak_in_use, P.in_use);
        return hipSuccess;
    }
    static const bool trace = getenv("DMRGX_POOL_TRACE") != nullptr;     // developer aid: driver allocations that miss the cache
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, c);
    if (trace && c >= ((size_t)8 << 20)) fprintf(stderr, "[pool] hipMalloc %.1f MiB: %.3f ms (cached %.1f GiB)\n", c / 1048576.0,
                                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), P.cached / 1073741824.0);
    if (e != hipSuccess) {                            // out of memory with blocks cached: release them and try once more
        (void)hipGetLastError();
        P.trim_locked(0);
        e = hipMalloc(out, c);
    }
    if (e == hipSuccess) { P.size_of[*out] = c; P.in_use += c; P.peak_in_use = std::max(P.peak_in_use, P.in_use); }
    else *out = nullptr;
    return e;
}

hipError_t pool_free(void* p)
{
    if (!p) return hipSuccess;
    Pool& P = pool();
    if (!P.enabled) return hipFree(p);
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) return hipFree(p);     // not ours
    P.free_by_size[it->second].push_back(p);
    P.freed_at[p] = ++P.clock;
    P.cached += it->second;
    P.in_use -= std::min(P.in_use, it->second);
    if (P.cached > P.cache_limit) P.trim_locked(P.cache_limit / 4 * 3);
    return hipSuccess;
}

// Host -> device copies of task tables and small operands: staged through a pinned ring so that the copy is a true
// asynchronous DMA (a hipMemcpyAsync from pageable memory blocks the host for ~20 us per call).  The ring is cut into
// H2D_SLOTS slots; every copy leaves an event on its stream in the slot it was staged in, and a slot is written again only
// after its events have completed (half a ring of uploads later they long have) -- no device-wide synchronisation.
namespace {
constexpr size_t H2D_RING_BYTES = (size_t)64 << 20, H2D_MAX_STAGED = (size_t)4 << 20, H2D_SLOTS = 16, H2D_SLOT_BYTES = H2D_RING_BYTES / H2D_SLOTS;
static_assert(H2D_MAX_STAGED <= H2D_SLOT_BYTES, "a staged copy fits one slot");
struct Ring {
    std::mutex mu; char* base = nullptr; size_t slot = 0, head = 0; bool failed = false;
    struct Pending { hipStream_t st; hipEvent_t ev; };
    std::vector<Pending> pending[H2D_SLOTS];          // one event per stream that copied out of the slot since it was opened
    std::vector<hipEvent_t> spare;
};
Ring& ring() { static Ring* r = new Ring(); return *r; }
}  // namespace

hipError_t h2d_async(void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    Ring& R = ring();
    if (bytes <= H2D_MAX_STAGED && !R.failed) {
        std::lock_guard<std::mutex> lock(R.mu);
        if (!R.base) { if (hipHostMalloc((void**)&R.base, H2D_RING_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); R.base = nullptr; R.failed = true; } }
        if (R.base) {
            const size_t need = (bytes + 255) & ~(size_t)255;
            if (R.head + need > H2D_SLOT_BYTES) {         // open the next slot: wait for the copies staged in it one lap ago
                R.slot = (R.slot + 1) % H2D_SLOTS; R.head = 0;
                for (Ring::Pending& p : R.pending[R.slot]) { hipError_t e = hipEventSynchronize(p.ev); if (e != hipSuccess) return e; R.spare.push_back(p.ev); }
                R.pending[R.slot].clear();
            }
            char* at = R.base + R.slot * H2D_SLOT_BYTES + R.head;
            R.head += need;
            memcpy(at, src, bytes);
            hipError_t e = hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return e;
            Ring::Pending* mine = nullptr;
            for (Ring::Pending& p : R.pending[R.slot]) if (p.st == st) mine = &p;
            if (!mine) {
                hipEvent_t ev;
                if (!R.spare.empty()) { ev = R.spare.back(); R.spare.pop_back(); }
                else { e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
                R.pending[R.slot].push_back(Ring::Pending{st, ev});
                mine = &R.pending[R.slot].back();
            }
            return hipEventRecord(mine->ev, st);          // re-recorded after every copy: covers the slot's last copy on this stream
        }
    }
    // large or unstaged: the source must outlive the copy and the callers drop it on return
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

void pool_stats(size_t* in_use, size_t* cached, size_t* peak)
{
    Pool& P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    if (in_use) *in_use = P.in_use;
    if (cached) *cached = P.cached;
    if (peak) *peak = P.peak_in_use;
}

hipError_t zero_async(void* p, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    if (((uintptr_t)p & 7) || (bytes & 7)) return hipMemsetAsync(p, 0, bytes, st);
    const size_t n = bytes / 8;
    const unsigned grid = (unsigned)std::min<size_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, st, (uint64_t*)p, n);
    return hipGetLastError();
}

}  // namespace dmrgx
