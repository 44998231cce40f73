/** @file petsc_compat.hpp
    Minimal stand-ins for the PETSc / SLEPc / MPI names that the DMRG.x operator API is written against
    (reference include/DMRGBlock.hpp, DMRGKron.hpp, DMRGBlockContainer.hpp, src/DMRG-SquareLattice.cpp), so that
    the driver compiles unchanged on top of the MI355X engine.  Nothing here computes: `Mat`/`Vec` are handles to
    device-resident sector-blocked data (SectorMat.hpp), the options database is a string map with PETSc syntax
    (`-key value`, comma lists, bare boolean flags), errors are integer codes propagated with CHKERRQ. */
#ifndef DMRGX_PETSC_COMPAT_HPP
#define DMRGX_PETSC_COMPAT_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <string>
#include <stdexcept>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include "dmrgx.h"

typedef int64_t PetscInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscErrorCode;
typedef int PetscMPIInt;
typedef double PetscLogDouble;
typedef enum { PETSC_FALSE = 0, PETSC_TRUE = 1 } PetscBool;
typedef int MPI_Comm;

#define PETSC_COMM_WORLD 1
#define PETSC_COMM_SELF 2
#define MPI_COMM_NULL 0
#define PETSC_DEFAULT (-2)
#define PETSC_DECIDE (-1)
#define PETSC_MAX_PATH_LEN 4096
#define PETSC_EXTERN extern "C"

/* error codes asserted by the reference's own tests (petscerror.h, PETSc 3.8) */
#define PETSC_ERR_MEM 55
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_CORRUPT 64
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_FILE_OPEN 65

#define LLD(INT) ((long long)(INT))
#define PetscUnlikely(x) (__builtin_expect(!!(x), 0))
#define PetscMin(a, b) (((a) < (b)) ? (a) : (b))
#define PetscMax(a, b) (((a) < (b)) ? (b) : (a))
#define PetscAbsScalar(a) (((a) < 0) ? -(a) : (a))

#define CHKERRQ(ierr) do { if (PetscUnlikely(ierr)) return (ierr); } while (0)
#define DMRGX_SETERR(code, ...) do { fprintf(stderr, "[dmrgx] %s:%d %s(): ", __FILE__, __LINE__, __func__); \
        fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return (code); } while (0)
#define SETERRQ(comm, code, msg) DMRGX_SETERR(code, "%s", msg)
#define SETERRQ1(comm, code, ...) DMRGX_SETERR(code, __VA_ARGS__)
#define SETERRQ2(comm, code, ...) DMRGX_SETERR(code, __VA_ARGS__)
#define SETERRQ3(comm, code, ...) DMRGX_SETERR(code, __VA_ARGS__)
#define SETERRQ4(comm, code, ...) DMRGX_SETERR(code, __VA_ARGS__)
#define SETERRQ5(comm, code, ...) DMRGX_SETERR(code, __VA_ARGS__)
#define CPP_CHKERR(ierr) do { if (ierr) fprintf(stderr, "[dmrgx] error %d in %s\n", (int)(ierr), __func__); } while (0)
#define CPP_CHKERRQ_MSG(ierr, msg) do { if (ierr) { fprintf(stderr, "[dmrgx] %s\n", msg); throw std::runtime_error(msg); } } while (0)

namespace dmrgx_host {

/** PETSc-style options database (process-wide, like PETSc's). */
class Options {
public:
    static Options& Global() { static Options o; return o; }
    void Clear() { kv.clear(); }
    void Set(const std::string& key, const std::string& val) { kv[key] = val; }
    void Parse(int argc, char** argv) {
        for (int i = 1; i < argc; ++i) {
            std::string a(argv[i]);
            if (a.size() < 2 || a[0] != '-' || IsNumber(a)) continue;
            std::string key = a.substr(1);
            if (i + 1 < argc && (argv[i + 1][0] != '-' || IsNumber(argv[i + 1]))) { kv[key] = argv[i + 1]; ++i; }
            else kv[key] = "";
        }
    }
    bool Find(const char* name, std::string& val) const {
        const char* n = (name[0] == '-') ? name + 1 : name;
        auto it = kv.find(n);
        if (it == kv.end()) return false;
        val = it->second;
        return true;
    }
    const std::map<std::string, std::string>& All() const { return kv; }
private:
    static bool IsNumber(const std::string& s) { char* e = nullptr; strtod(s.c_str(), &e); return e && *e == '\0' && !s.empty(); }
    std::map<std::string, std::string> kv;
};

inline int& WorldSize() { static int v = 1; return v; }
inline int& WorldRank() { static int v = 0; return v; }
/** the process-wide communicator (NULL on one rank): RCCL over xGMI, or the host-staged rehearsal back-end */
inline dmrgx_comm*& WorldComm() { static dmrgx_comm* c = nullptr; return c; }

/** One process per GPU, started by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run's
    convention; `mpirun`-style launchers of the reference map onto it with a two-line wrapper).  Takes the place of
    MPI_Init inside SlepcInitialize: selects the device, creates the communicator.
      DMRGX_COMM=rccl (default)  RCCL; the 128-byte id goes from rank 0 to the others through a rendezvous file
                                 (DMRGX_RDZV_FILE, default /tmp/dmrgx_rdzv_<launcher pid>_<MASTER_PORT>)
      DMRGX_COMM=shm             all ranks share one GPU and exchange through POSIX shared memory DMRGX_SHM_NAME
                                 (rehearsal of the multi-rank control flow on a one-GPU box; tests) */
inline int CommBootstrap()
{
    const char* ws = getenv("WORLD_SIZE");
    const int world = ws ? atoi(ws) : 1;
    if (world <= 1 && !getenv("DMRGX_FORCE_COMM")) return 0;      /* DMRGX_FORCE_COMM=1: a one-rank communicator (tests the RCCL start-up path on one GPU) */
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
    const int local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
    const std::string mode = getenv("DMRGX_COMM") ? getenv("DMRGX_COMM") : "rccl";
    /* per-launch tag of the rendezvous names: launcher pid + port + the launcher's run id / restart count where it exports one
       (torch.distributed.run does), so that a restarted worker group never reads the id of the previous attempt */
    const std::string tag = std::to_string((long)getppid()) + "_" + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") +
                            (getenv("TORCHELASTIC_RUN_ID") ? std::string("_") + getenv("TORCHELASTIC_RUN_ID") : std::string()) +
                            (getenv("TORCHELASTIC_RESTART_COUNT") ? std::string("_") + getenv("TORCHELASTIC_RESTART_COUNT") : std::string());
    dmrgx_comm* comm = nullptr;
    if (mode == "shm") {
        if (dmrgx_set_device(0)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
        const std::string name = getenv("DMRGX_SHM_NAME") ? getenv("DMRGX_SHM_NAME") : "dmrgx_shm_" + tag;
        if (dmrgx_comm_init_host_staged(rank, world, name.c_str(), &comm)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
    } else {
        if (dmrgx_set_device(local)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
        const std::string path = getenv("DMRGX_RDZV_FILE") ? getenv("DMRGX_RDZV_FILE") : "/tmp/dmrgx_rdzv_" + tag;
        uint8_t id[DMRGX_COMM_ID_BYTES];
        int ndev = 0;
        if (dmrgx_device_count(&ndev) == 0 && world > ndev && !getenv("DMRGX_MULTI_NODE")) {
            fprintf(stderr, "[dmrgx] WORLD_SIZE %d exceeds the %d GPU(s) of this node (the rendezvous file is node-local)\n", world, ndev); return 1; }
        /* The file carries the launch it belongs to: a 64-byte nonce in front of the id -- DMRGX_LAUNCH_NONCE where the launcher exports
           one (bench.py does), else the per-launch tag above (launcher pid, port, run id, restart count: the same on every rank of one
           launch, different for the next).  The other ranks wait for a file with THEIR nonce, so the leftover of a run that died before
           rank 0 removed it is never taken for this launch's, however recent it is and whatever the clocks say (the
           round-3 guard compared the file's mtime with the local clock). */
        char nonce[64];
        memset(nonce, 0, sizeof(nonce));
        strncpy(nonce, getenv("DMRGX_LAUNCH_NONCE") ? getenv("DMRGX_LAUNCH_NONCE") : tag.c_str(), sizeof(nonce) - 1);
        if (rank == 0) {
            if (dmrgx_comm_unique_id(id)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
            const std::string tmp = path + ".tmp";
            unlink(path.c_str());                                  /* a file left behind by a run that died before its barrier */
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(nonce, 1, sizeof(nonce), f) != sizeof(nonce) || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "[dmrgx] cannot write %s\n", tmp.c_str()); return 1; }
            fclose(f);
            if (rename(tmp.c_str(), path.c_str()) != 0) { fprintf(stderr, "[dmrgx] cannot publish %s\n", path.c_str()); return 1; }
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            while (true) {
                char seen[64];
                FILE* f = fopen(path.c_str(), "rb");
                if (f) {
                    const size_t n1 = fread(seen, 1, sizeof(seen), f), n2 = fread(id, 1, sizeof(id), f);
                    fclose(f);
                    if (n1 == sizeof(seen) && n2 == sizeof(id) && memcmp(seen, nonce, sizeof(nonce)) == 0) break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300)) { fprintf(stderr, "[dmrgx] rank %d: no rendezvous file %s of this launch\n", rank, path.c_str()); return 1; }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        if (dmrgx_comm_init(rank, world, id, &comm)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
        if (dmrgx_comm_barrier(comm, nullptr)) { fprintf(stderr, "[dmrgx] %s\n", dmrgx_last_error()); return 1; }
        if (rank == 0) unlink(path.c_str());
    }
    WorldSize() = world; WorldRank() = rank; WorldComm() = comm;
    return 0;
}

}  // namespace dmrgx_host

inline PetscErrorCode PetscOptionsGetInt(void*, void*, const char* name, PetscInt* v, PetscBool* set) {
    std::string s; const bool f = dmrgx_host::Options::Global().Find(name, s);
    if (f && !s.empty()) *v = (PetscInt)strtoll(s.c_str(), nullptr, 10);
    if (set) *set = f ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
inline PetscErrorCode PetscOptionsGetReal(void*, void*, const char* name, PetscReal* v, PetscBool* set) {
    std::string s; const bool f = dmrgx_host::Options::Global().Find(name, s);
    if (f && !s.empty()) *v = strtod(s.c_str(), nullptr);
    if (set) *set = f ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
inline PetscErrorCode PetscOptionsGetBool(void*, void*, const char* name, PetscBool* v, PetscBool* set) {
    std::string s; const bool f = dmrgx_host::Options::Global().Find(name, s);
    if (f) *v = (s.empty() || s == "1" || s == "true" || s == "yes" || s == "TRUE" || s == "on") ? PETSC_TRUE : PETSC_FALSE;
    if (set) *set = f ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
inline PetscErrorCode PetscOptionsGetString(void*, void*, const char* name, char* buf, size_t len, PetscBool* set) {
    std::string s; const bool f = dmrgx_host::Options::Global().Find(name, s);
    if (f) { strncpy(buf, s.c_str(), len - 1); buf[len - 1] = 0; }
    if (set) *set = f ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
inline PetscErrorCode PetscOptionsGetIntArray(void*, void*, const char* name, PetscInt* arr, PetscInt* n, PetscBool* set) {
    std::string s; const bool f = dmrgx_host::Options::Global().Find(name, s);
    PetscInt cnt = 0;
    if (f) {
        size_t pos = 0;
        while (pos <= s.size() && cnt < *n) {
            size_t c = s.find(',', pos);
            std::string tok = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
            if (!tok.empty()) arr[cnt++] = (PetscInt)strtoll(tok.c_str(), nullptr, 10);
            if (c == std::string::npos) break;
            pos = c + 1;
        }
    }
    *n = cnt;
    if (set) *set = f ? PETSC_TRUE : PETSC_FALSE;
    return 0;
}
inline PetscErrorCode PetscOptionsSetValue(void*, const char* name, const char* value) {
    dmrgx_host::Options::Global().Set(name[0] == '-' ? name + 1 : name, value ? value : "");
    return 0;
}

inline PetscErrorCode MPI_Comm_size(MPI_Comm, PetscMPIInt* n) { *n = dmrgx_host::WorldSize(); return 0; }
inline PetscErrorCode MPI_Comm_rank(MPI_Comm, PetscMPIInt* r) { *r = dmrgx_host::WorldRank(); return 0; }
inline PetscErrorCode MPI_Barrier(MPI_Comm) { return dmrgx_host::WorldComm() ? (PetscErrorCode)dmrgx_comm_barrier(dmrgx_host::WorldComm(), nullptr) : 0; }

inline PetscErrorCode SlepcInitialize(int* argc, char*** argv, const char*, const char*) {
    if (argc && argv) dmrgx_host::Options::Global().Parse(*argc, *argv);
    return dmrgx_host::CommBootstrap();
}
inline PetscErrorCode SlepcFinalize() {
    if (dmrgx_host::WorldComm()) { dmrgx_comm_destroy(dmrgx_host::WorldComm()); dmrgx_host::WorldComm() = nullptr; }
    return 0;
}
inline PetscErrorCode PetscTime(PetscLogDouble* t) {
    *t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return 0;
}
#define PetscPrintf(comm, ...) (printf(__VA_ARGS__), 0)
inline PetscErrorCode PetscFOpen(MPI_Comm, const char* fn, const char* mode, FILE** fp) {
    /* like PETSc's: only the first rank of the communicator owns the file; the others write into the void */
    *fp = fopen(dmrgx_host::WorldRank() == 0 ? fn : "/dev/null", mode);
    if (!*fp) { fprintf(stderr, "[dmrgx] cannot open %s\n", fn); return PETSC_ERR_FILE_OPEN; }
    return 0;
}
inline PetscErrorCode PetscFClose(MPI_Comm, FILE* fp) { if (fp) fclose(fp); return 0; }
inline PetscErrorCode PetscTestDirectory(const char* d, char, PetscBool* flg) {
    struct stat st; *flg = (stat(d, &st) == 0 && S_ISDIR(st.st_mode)) ? PETSC_TRUE : PETSC_FALSE; return 0;
}
inline PetscErrorCode PetscTestFile(const char* d, char, PetscBool* flg) {
    struct stat st; *flg = (stat(d, &st) == 0 && S_ISREG(st.st_mode)) ? PETSC_TRUE : PETSC_FALSE; return 0;
}
inline PetscErrorCode Makedir(const std::string& dir) {
    std::string cur;
    for (size_t i = 0; i < dir.size(); ++i) { cur += dir[i]; if (dir[i] == '/' || i + 1 == dir.size()) mkdir(cur.c_str(), 0777); }
    return 0;
}

#endif
