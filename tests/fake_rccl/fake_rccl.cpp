// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl that carries the collectives of the library's multi-GPU path
// between PROCESSES THAT SHARE ONE GPU (RCCL itself refuses two ranks on one device, and the test boxes have one GPU).
// Loaded through QGD_RCCL_LIB (csrc/qgd_host_comm.cpp: load_rccl), it exports the seven entry points the library binds.  Data
// travel through a POSIX shared-memory segment named after the unique id: every collective synchronises the caller's
// stream, copies the rank's contribution to its slot, meets the other ranks at a barrier, assembles the result on the
// device and meets them again.  Synchronous and slow on purpose; what it exercises is the PRODUCT's side of the protocol in
// C++ with world > 1 -- comm_discrete_adjoint / comm_eval_forward: in-place all-gather offsets, the out-of-place reduction,
// the deferred terminal condition on the last rank, the zeroed scalars of column ranks != 0 -- in separate processes.
// A rank that does not arrive within FAKE_RCCL_TIMEOUT_MS (default 20000) makes the others return ncclSystemError.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;       // the library passes ncclDouble (7 in nccl.h) only
typedef int ncclRedOp_t;          // ncclSum (0) only

struct Shared {
    std::atomic<int> arrived, generation, attached, failed;
    size_t slot_bytes;
    int nranks;
};
struct FakeComm {
    int rank, nranks;
    Shared *sh;
    char *slots;
    size_t map_bytes;
    std::string name;
};
typedef FakeComm *ncclComm_t;

static const size_t SLOT = (size_t)48 << 20;      // per rank: C5's window products are 2 MB, its state panels 1 MB

static double timeout_ms() { const char *e = getenv("FAKE_RCCL_TIMEOUT_MS"); return e ? atof(e) : 20000.0; }

static bool barrier(FakeComm *c)
{
    Shared *s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->nranks) { s->arrived.store(0); s->generation.fetch_add(1); return true; }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (s->failed.load()) return false;
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) { s->failed.store(1); return false; }
        sched_yield();
    }
    return true;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    unsigned long long r[2] = {(unsigned long long)getpid() * 2654435761ull, (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count()};
    if (FILE *f = fopen("/dev/urandom", "rb")) { if (fread(r, sizeof r, 1, f) != 1) { /* keep the fallback */ } fclose(f); }
    snprintf(id->internal, sizeof id->internal, "/qgd_fake_rccl_%016llx%016llx", r[0], r[1]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    const size_t bytes = 4096 + SLOT * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            fd = shm_open(id.internal, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) return ncclSystemError;
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    FakeComm *c = new FakeComm{rank, nranks, (Shared *)p, (char *)p + 4096, bytes, id.internal};
    if (rank == 0) { c->sh->slot_bytes = SLOT; c->sh->nranks = nranks; }
    c->sh->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < nranks) {       // ncclCommInitRank is collective
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) { munmap(p, bytes); delete c; return ncclSystemError; }
        usleep(500);
    }
    *out = c;
    return ncclSuccess;
}

static ncclResult_t finish(FakeComm *c)
{
    munmap((void *)c->sh, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { return c ? finish(c) : ncclInvalidArgument; }
ncclResult_t ncclCommAbort(ncclComm_t c) { if (!c) return ncclInvalidArgument; c->sh->failed.store(1); return finish(c); }
ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *e) { *e = (c && c->sh->failed.load()) ? ncclRemoteError : ncclSuccess; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclSystemError ? "fake_rccl: a rank did not arrive (system error)" : "fake_rccl error"; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * sizeof(double);
    if (bytes > SLOT) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + SLOT * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->nranks; r++)
        if (hipMemcpy((char *)recv + bytes * r, c->slots + SLOT * r, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * sizeof(double);
    if (bytes > SLOT) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + SLOT * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < c->nranks; r++) {           // rank order: every rank gets the same bits
        const double *s = (const double *)(c->slots + SLOT * r);
        for (size_t i = 0; i < count; i++) sum[i] += s[i];
    }
    if (hipMemcpy(recv, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

}  // extern "C"
