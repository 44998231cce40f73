// Internal helpers shared by the HIP translation units of libdmrgx_hip.so (not part of the ABI).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#include "dmrgx.h"

namespace dmrgx {

void set_error(const char* fmt, ...);

#define DMRGX_HIP(call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            ::dmrgx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return DMRGX_ERR_DEVICE;                                                                 \
        }                                                                                            \
    } while (0)

#define DMRGX_CHK(call)                    \
    do {                                   \
        dmrgx_status s__ = (call);         \
        if (s__ != DMRGX_OK) return s__;   \
    } while (0)

#define DMRGX_FAIL(code, ...)              \
    do {                                   \
        ::dmrgx::set_error(__VA_ARGS__);   \
        return (code);                     \
    } while (0)

// Stripe rule of the multi-GPU layout: rank w owns columns [stripe_cut(n,W,w), stripe_cut(n,W,w+1)) of a KronBlock.
// Cuts sit at the even split rounded to a multiple of the MFMA block width (16 columns) -- every stripe is within 16 columns of
// n / W, so no rank carries the whole remainder (a cut on whole 64-column GEMM tiles with the remainder on the last rank gave
// 64,128,..,168 for n = 1000, W = 8: the slowest rank 34 % above the even share) -- and snap to a multiple of the GEMM tile
// width (64) when that is no further from the even split than 16 columns, which keeps a rank from paying for an almost empty tile
// column per operator where it costs no balance; below 48 W columns the plain even split is used.
inline int32_t stripe_cut(int32_t n, int32_t W, int32_t w)
{
    if (w <= 0) return 0;
    if (w >= W) return n;
    const int64_t even = ((int64_t)n * w) / W;
    if ((int64_t)n < (int64_t)48 * W) return (int32_t)even;      // (every cut moves by at most 16 columns: stripes of >= 48 keep >= 16)
    const int64_t c16 = ((even + 8) / 16) * 16, c64 = ((even + 32) / 64) * 64;
    const int64_t d64 = c64 > even ? c64 - even : even - c64;
    const int64_t cut = (d64 <= 16 && c64 > 0 && c64 < n) ? c64 : c16;
    return (int32_t)std::min<int64_t>(std::max<int64_t>(cut, 0), n);
}

// Pooled device memory and a fill kernel (pool.hip): pool_free never synchronises, recycling is stream-ordered.
hipError_t pool_malloc(void** out, size_t bytes);
hipError_t pool_free(void* p);
hipError_t zero_async(void* p, size_t bytes, hipStream_t st);
void pool_stats(size_t* in_use, size_t* cached, size_t* peak);
hipError_t h2d_async(void* dst, const void* src, size_t bytes, hipStream_t st);   // pinned-ring staged, never blocks on the copy

// Which stripe of KronBlock k rank w owns: the stripes are dealt round the ranks block by block, so that the rank holding the
// last stripe of a block -- the one with the ragged remainder of n, i.e. an extra, almost empty tile column -- changes from block
// to block instead of always being rank W-1 (measured on one GPU, W = 8: slowest rank 0.66 ms against 0.44 ms for the fastest).
inline int32_t stripe_of_rank(int32_t W, int32_t w, int32_t k) { return (w + k) % W; }

// RAII device buffer owned by a plan.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    dmrgx_status alloc(size_t n) {
        release();
        if (n == 0) return DMRGX_OK;
        hipError_t e = pool_malloc(&p, n);
        if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); return DMRGX_ERR_MEM; }
        bytes = n;
        return DMRGX_OK;
    }
    void release() { if (p && owned) (void)pool_free(p); p = nullptr; bytes = 0; owned = true; }
    // n bytes at `ptr` inside another buffer (a table of a PackedUpload): same accessors, nothing to free; the arena must outlive it
    void view(void* ptr, size_t n) { release(); p = n ? ptr : nullptr; bytes = n; owned = false; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    bool owned = true;
};

template <class T>
dmrgx_status upload(DevBuf& buf, const std::vector<T>& host, hipStream_t st) {
    DMRGX_CHK(buf.alloc(host.size() * sizeof(T)));
    if (!host.empty()) DMRGX_HIP(h2d_async(buf.p, host.data(), host.size() * sizeof(T), st));
    return DMRGX_OK;
}

// Several task tables in ONE host -> device copy: every copy is a stream operation of its own (~4 us of blit kernel behind a ~6 us
// gap at small m, where a step used to issue ~60 of them); the tables of one call go out together and are addressed by offset.
struct PackedUpload {
    std::vector<char> host;
    template <class T> size_t add(const std::vector<T>& v) {
        const size_t off = (host.size() + 15) & ~(size_t)15, bytes = v.size() * sizeof(T);
        if (host.capacity() < off + bytes) host.reserve(std::max(2 * host.capacity(), off + bytes + (bytes >> 1)));
        host.resize(off);                                                       // (alignment padding only: the table itself is appended, not zero-filled first)
        if (bytes) { const char* p = reinterpret_cast<const char*>(v.data()); host.insert(host.end(), p, p + bytes); }
        return off;
    }
    dmrgx_status upload(DevBuf& buf, hipStream_t st) {
        DMRGX_CHK(buf.alloc(std::max<size_t>(host.size(), 16)));
        // (in pieces the pinned ring stages -- 4 MiB: a larger copy would leave pageable memory synchronously; the plan tables of an
        //  m = 4096 step are ~6 MB together)
        constexpr size_t piece = (size_t)4 << 20;
        for (size_t off = 0; off < host.size(); off += piece)
            DMRGX_HIP(h2d_async(static_cast<char*>(buf.p) + off, host.data() + off, std::min(piece, host.size() - off), st));
        return DMRGX_OK;
    }
    // `dst` becomes a view of the table added at `off` (n elements of T) inside the uploaded arena
    template <class T> static void view(DevBuf& dst, const DevBuf& arena, size_t off, size_t n) { dst.view(static_cast<char*>(arena.p) + off, n * sizeof(T)); }
};
template <class T> inline T* packed_at(const DevBuf& buf, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(buf.p) + off); }

}  // namespace dmrgx
