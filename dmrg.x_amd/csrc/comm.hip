// Communicator of the multi-GPU path: one process per GPU, collectives over RCCL (xGMI) issued from C++ on the caller's
// stream -- no Python, no host hop in the Lanczos step.
//
// What it replaces in the reference (MPI on CPU): the VecScatter-to-all of x at the top of every MatMult
// (src/DMRGKron.cpp:1833-1834) -> dmrgx_comm_allgather of the Krylov vector's rank segments; the MPI_Allreduce behind SLEPc's
// VecDot / VecNorm -> ONE fused dmrgx_comm_allreduce_sum of <= ncv + 1 doubles per Gram-Schmidt pass; the rank-0 RDM solve +
// broadcast of the rotation (include/DMRGBlockContainer.hpp:1673-1677, 1812-1925) -> density matrices dealt over the ranks,
// spectra exchanged with dmrgx_comm_allgather_host, kept eigenvectors sent with dmrgx_comm_bcast.
//
// Two back-ends behind the same entry points:
//   RCCL        -- the product path.  librccl.so (570 MB) is opened at dmrgx_comm_init, not at library load: single-GPU runs never
//                  touch it.  In-place ncclAllGather / ncclAllReduce / ncclBroadcast on the caller's stream.
//   host-staged -- several ranks sharing ONE GPU exchange through a POSIX shared-memory segment (RCCL refuses two ranks on one
//                  device).  Rehearsal of the N > 1 control flow on a one-GPU box (tests); never used for measurements.
#include "common.h"
#include "symeig.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <thread>

namespace dmrgx {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// librccl.so is resolved once per process, on first use
dmrgx_status rccl_api(const RcclApi** out)
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle) {
            auto sym = [&](const char* n) { return dlsym(api.handle, n); };
            api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
            api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
            api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
            api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        }
    }
    if (!api.handle) DMRGX_FAIL(DMRGX_ERR_DEVICE, "comm: cannot open librccl.so (%s)", dlerror());
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.Broadcast || !api.GetErrorString)
        DMRGX_FAIL(DMRGX_ERR_DEVICE, "comm: librccl.so lacks a required entry point");
    *out = &api;
    return DMRGX_OK;
}

#define DMRGX_NCCL(api, call)                                                                          \
    do {                                                                                               \
        ncclResult_t r__ = (call);                                                                     \
        if (r__ != ncclSuccess) DMRGX_FAIL(DMRGX_ERR_DEVICE, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, (api)->GetErrorString(r__)); \
    } while (0)

// ---- host-staged back-end: shared segment = header + world slots --------------------------------------------------
struct ShmHeader {
    uint32_t magic;                 // set last by rank 0
    int32_t world;
    uint64_t slot_bytes;
    pthread_barrier_t barrier;      // process-shared
};
constexpr uint32_t SHM_MAGIC = 0x444d5258u;

}  // namespace
}  // namespace dmrgx

using namespace dmrgx;

struct dmrgx_comm {
    int32_t rank = 0, world = 1;
    int32_t backend = 0;            // DMRGX_COMM_RCCL | DMRGX_COMM_HOST_STAGED
    const RcclApi* api = nullptr;
    ncclComm_t nccl = nullptr;
    DevBuf scratch;                 // staging of host payloads for the RCCL back-end
    // host-staged
    std::string shm_name;
    ShmHeader* hdr = nullptr;
    char* slots = nullptr;
    size_t map_bytes = 0, slot_bytes = 0;
    std::vector<char> host;         // pageable staging buffer
    char* pinned = nullptr;         // RCCL back-end: pinned staging of small host payloads (dmrgx_comm_allgather_host), grown on demand
    size_t pinned_bytes = 0;
};

extern "C" dmrgx_status dmrgx_set_device(int32_t device)
{
    int n = 0;
    DMRGX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) DMRGX_FAIL(DMRGX_ERR_ARG, "set_device: device %d of %d", device, n);
    DMRGX_HIP(hipSetDevice(device));
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_unique_id(uint8_t* id)
{
    if (!id) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_unique_id: null argument");
    const RcclApi* api = nullptr;
    DMRGX_CHK(rccl_api(&api));
    static_assert(sizeof(ncclUniqueId) == DMRGX_COMM_ID_BYTES, "unique id size");
    ncclUniqueId u;
    DMRGX_NCCL(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_init(int32_t rank, int32_t world, const uint8_t* id, dmrgx_comm** out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_init: bad argument (rank %d of %d)", rank, world);
    *out = nullptr;
    const RcclApi* api = nullptr;
    DMRGX_CHK(rccl_api(&api));
    std::unique_ptr<dmrgx_comm> C(new (std::nothrow) dmrgx_comm());
    if (!C) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
    C->rank = rank; C->world = world; C->backend = DMRGX_COMM_RCCL; C->api = api;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    DMRGX_NCCL(api, api->CommInitRank(&C->nccl, world, u, rank));
    *out = C.release();
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_init_host_staged(int32_t rank, int32_t world, const char* shm_name, dmrgx_comm** out)
{
    if (!out || !shm_name || !shm_name[0] || world < 1 || rank < 0 || rank >= world) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_init_host_staged: bad argument");
    *out = nullptr;
    std::unique_ptr<dmrgx_comm> C(new (std::nothrow) dmrgx_comm());
    if (!C) DMRGX_FAIL(DMRGX_ERR_MEM, "out of host memory");
    C->rank = rank; C->world = world; C->backend = DMRGX_COMM_HOST_STAGED;
    if (world > 1) symeig_set_persistent(false);      // several ranks share ONE GPU here: persistent kernels that each want most CUs would wait on each other
    C->shm_name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
    const size_t total_mb = getenv("DMRGX_SHM_MB") ? (size_t)atol(getenv("DMRGX_SHM_MB")) : 256;
    const size_t hdr_bytes = 4096;
    C->slot_bytes = ((total_mb << 20) / (size_t)world) & ~(size_t)4095;
    C->map_bytes = hdr_bytes + C->slot_bytes * (size_t)world;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(C->shm_name.c_str());
        fd = shm_open(C->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)C->map_bytes) != 0) { if (fd >= 0) close(fd); DMRGX_FAIL(DMRGX_ERR_DEVICE, "comm: cannot create shared segment %s", C->shm_name.c_str()); }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (true) {
            fd = shm_open(C->shm_name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= C->map_bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) DMRGX_FAIL(DMRGX_ERR_DEVICE, "comm: rank 0 never created %s", C->shm_name.c_str());
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    void* m = mmap(nullptr, C->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) DMRGX_FAIL(DMRGX_ERR_MEM, "comm: mmap of %s failed", C->shm_name.c_str());
    C->hdr = (ShmHeader*)m;
    C->slots = (char*)m + hdr_bytes;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&C->hdr->barrier, &a, (unsigned)world);
        pthread_barrierattr_destroy(&a);
        C->hdr->world = world; C->hdr->slot_bytes = C->slot_bytes;
        __atomic_store_n(&C->hdr->magic, SHM_MAGIC, __ATOMIC_RELEASE);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(&C->hdr->magic, __ATOMIC_ACQUIRE) != SHM_MAGIC) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) DMRGX_FAIL(DMRGX_ERR_DEVICE, "comm: shared segment %s never initialised", C->shm_name.c_str());
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (C->hdr->world != world || C->hdr->slot_bytes != C->slot_bytes) DMRGX_FAIL(DMRGX_ERR_ARG, "comm: %s belongs to a run with another world size", C->shm_name.c_str());
    }
    pthread_barrier_wait(&C->hdr->barrier);
    if (rank == 0) shm_unlink(C->shm_name.c_str());      // everyone has it mapped: the name can go
    *out = C.release();
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_info(const dmrgx_comm* C, int32_t* rank, int32_t* world, int32_t* backend)
{
    if (!C) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_info: null communicator");
    if (rank) *rank = C->rank;
    if (world) *world = C->world;
    if (backend) *backend = C->backend;
    return DMRGX_OK;
}

namespace {
inline void shm_barrier(dmrgx_comm* C) { pthread_barrier_wait(&C->hdr->barrier); }

// every rank contributes `bytes` device bytes at src; rank p's contribution lands at dst_of(p) on every rank
template <class DstOf>
dmrgx_status staged_allgather(dmrgx_comm* C, const char* src, size_t bytes, DstOf dst_of, bool skip_self, hipStream_t st)
{
    DMRGX_HIP(hipStreamSynchronize(st));
    for (size_t o = 0; o < bytes; o += C->slot_bytes) {
        const size_t n = std::min(C->slot_bytes, bytes - o);
        DMRGX_HIP(hipMemcpy(C->slots + (size_t)C->rank * C->slot_bytes, src + o, n, hipMemcpyDeviceToHost));
        shm_barrier(C);
        for (int p = 0; p < C->world; ++p) {
            if (skip_self && p == C->rank) continue;
            DMRGX_HIP(hipMemcpy(dst_of(p) + o, C->slots + (size_t)p * C->slot_bytes, n, hipMemcpyHostToDevice));
        }
        shm_barrier(C);
    }
    return DMRGX_OK;
}
}  // namespace

extern "C" dmrgx_status dmrgx_comm_allgather(dmrgx_comm* C, double* full, int64_t seg_stride, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!C || !full || seg_stride < 0) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_allgather: bad argument");
    if (C->world == 1 || seg_stride == 0) return DMRGX_OK;
    if (C->backend == DMRGX_COMM_RCCL) {
        DMRGX_NCCL(C->api, C->api->AllGather(full + (size_t)C->rank * seg_stride, full, (size_t)seg_stride, ncclDouble, C->nccl, st));   // in place
        return DMRGX_OK;
    }
    char* base = (char*)full;
    const size_t seg = (size_t)seg_stride * sizeof(double);
    return staged_allgather(C, base + (size_t)C->rank * seg, seg, [&](int p) { return base + (size_t)p * seg; }, true, st);
}

extern "C" dmrgx_status dmrgx_comm_allreduce_sum(dmrgx_comm* C, double* buf, int64_t count, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!C || count < 0 || (count > 0 && !buf)) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_allreduce_sum: bad argument");
    if (C->world == 1 || count == 0) return DMRGX_OK;
    if (C->backend == DMRGX_COMM_RCCL) {
        DMRGX_NCCL(C->api, C->api->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, C->nccl, st));
        return DMRGX_OK;
    }
    const size_t bytes = (size_t)count * sizeof(double);
    if (bytes > C->slot_bytes) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_allreduce_sum: %zu bytes exceed the staging slot (DMRGX_SHM_MB)", bytes);
    DMRGX_HIP(hipStreamSynchronize(st));
    DMRGX_HIP(hipMemcpy(C->slots + (size_t)C->rank * C->slot_bytes, buf, bytes, hipMemcpyDeviceToHost));
    shm_barrier(C);
    std::vector<double> sum((size_t)count, 0.0);
    for (int p = 0; p < C->world; ++p) {                     // rank order: every rank forms the same sum, bit for bit
        const double* s = (const double*)(C->slots + (size_t)p * C->slot_bytes);
        for (int64_t i = 0; i < count; ++i) sum[(size_t)i] += s[i];
    }
    DMRGX_HIP(hipMemcpy(buf, sum.data(), bytes, hipMemcpyHostToDevice));
    shm_barrier(C);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_bcast(dmrgx_comm* C, void* buf, size_t bytes, int32_t root, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!C || root < 0 || root >= C->world || (bytes > 0 && !buf)) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_bcast: bad argument");
    if (C->world == 1 || bytes == 0) return DMRGX_OK;
    if (C->backend == DMRGX_COMM_RCCL) {
        DMRGX_NCCL(C->api, C->api->Broadcast(buf, buf, bytes, ncclChar, root, C->nccl, st));
        return DMRGX_OK;
    }
    DMRGX_HIP(hipStreamSynchronize(st));
    const size_t cap = C->slot_bytes * (size_t)C->world;     // the whole data area serves as one buffer
    for (size_t o = 0; o < bytes; o += cap) {
        const size_t n = std::min(cap, bytes - o);
        if (C->rank == root) DMRGX_HIP(hipMemcpy(C->slots, (char*)buf + o, n, hipMemcpyDeviceToHost));
        shm_barrier(C);
        if (C->rank != root) DMRGX_HIP(hipMemcpy((char*)buf + o, C->slots, n, hipMemcpyHostToDevice));
        shm_barrier(C);
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_allgather_host(dmrgx_comm* C, const void* send, void* recv, size_t bytes_per_rank, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!C || (bytes_per_rank > 0 && (!send || !recv))) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_allgather_host: bad argument");
    if (bytes_per_rank == 0) return DMRGX_OK;
    if (C->world == 1) { memmove(recv, send, bytes_per_rank); return DMRGX_OK; }
    if (C->backend == DMRGX_COMM_RCCL) {
        const size_t seg = (bytes_per_rank + 7) & ~(size_t)7, need = seg * (size_t)C->world;
        if (C->scratch.bytes < need) DMRGX_CHK(C->scratch.alloc(need));
        char* d = C->scratch.as<char>();
        // host -> pinned -> device, all-gather on the device, device -> pinned -> host: three asynchronous operations and ONE wait (copies
        // from and to pageable memory are staged by the runtime and block the host once each: two extra stalls per truncation step)
        if (C->pinned_bytes < need + seg) {
            if (C->pinned) { DMRGX_HIP(hipStreamSynchronize(st)); (void)hipHostFree(C->pinned); C->pinned = nullptr; C->pinned_bytes = 0; }
            DMRGX_HIP(hipHostMalloc((void**)&C->pinned, 2 * (need + seg), hipHostMallocDefault));
            C->pinned_bytes = 2 * (need + seg);
        }
        char* hs = C->pinned;              // this rank's segment going out
        char* hr = C->pinned + seg;        // everybody's segments coming back
        memcpy(hs, send, bytes_per_rank);
        DMRGX_HIP(hipMemcpyAsync(d + (size_t)C->rank * seg, hs, bytes_per_rank, hipMemcpyHostToDevice, st));
        DMRGX_NCCL(C->api, C->api->AllGather(d + (size_t)C->rank * seg, d, seg, ncclChar, C->nccl, st));
        DMRGX_HIP(hipMemcpyAsync(hr, d, need, hipMemcpyDeviceToHost, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        for (int p = 0; p < C->world; ++p) memcpy((char*)recv + (size_t)p * bytes_per_rank, hr + (size_t)p * seg, bytes_per_rank);
        return DMRGX_OK;
    }
    DMRGX_HIP(hipStreamSynchronize(st));
    for (size_t o = 0; o < bytes_per_rank; o += C->slot_bytes) {
        const size_t n = std::min(C->slot_bytes, bytes_per_rank - o);
        memcpy(C->slots + (size_t)C->rank * C->slot_bytes, (const char*)send + o, n);
        shm_barrier(C);
        for (int p = 0; p < C->world; ++p) memcpy((char*)recv + (size_t)p * bytes_per_rank + o, C->slots + (size_t)p * C->slot_bytes, n);
        shm_barrier(C);
    }
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_barrier(dmrgx_comm* C, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!C) DMRGX_FAIL(DMRGX_ERR_ARG, "comm_barrier: null communicator");
    if (C->world == 1) { DMRGX_HIP(hipStreamSynchronize(st)); return DMRGX_OK; }
    if (C->backend == DMRGX_COMM_RCCL) {
        if (C->scratch.bytes < 64) DMRGX_CHK(C->scratch.alloc(4096));
        DMRGX_HIP(zero_async(C->scratch.p, 8, st));
        DMRGX_NCCL(C->api, C->api->AllReduce(C->scratch.p, C->scratch.p, 1, ncclDouble, ncclSum, C->nccl, st));
        DMRGX_HIP(hipStreamSynchronize(st));
        return DMRGX_OK;
    }
    DMRGX_HIP(hipStreamSynchronize(st));
    shm_barrier(C);
    return DMRGX_OK;
}

extern "C" dmrgx_status dmrgx_comm_destroy(dmrgx_comm* C)
{
    if (!C) return DMRGX_OK;
    if (C->backend == DMRGX_COMM_RCCL && C->nccl) { (void)hipDeviceSynchronize(); (void)C->api->CommDestroy(C->nccl); }
    if (C->hdr) munmap((void*)C->hdr, C->map_bytes);
    if (C->pinned) (void)hipHostFree(C->pinned);
    delete C;
    return DMRGX_OK;
}
