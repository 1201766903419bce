// TEST TRANSPORT for csrc/comm.hip, compiled in by -DBDR_COMM_HOST_TRANSPORT (border_amd/build.py builds it as a SECOND library,
// libborder_amd_hostcomm.so; the product library never contains it and nothing selects it at run time).
//
// Why it exists: RCCL refuses two ranks on one device, so on a 1-GPU box everything in comm.hip that only happens with nranks > 1 - the
// agreement before a collective, the per-segment overlapped exchange meeting a second rank, the 1/N scale, gradient all-reduce in
// synchronous data-parallel mode, the staleness of derived weight copies after an exchange - could never execute.
// This file implements the six librccl entry points comm.hip binds (same signatures, same call sequence above them: bdr_comm_*,
// bdr_agent_allreduce_params, bdr_agent_broadcast_params, bdr_agent_set_grad_comm are compiled from the SAME source) over a POSIX
// shared-memory segment between processes of one host, so that N ranks may share one GPU:
//   all-reduce = stream sync -> device-to-host into this rank's slot -> barrier -> every rank reduces the slots in RANK ORDER (the same
//   f32 sum on every rank) -> barrier -> host-to-device on the caller's stream -> stream sync.
// It is a correctness vehicle: blocking, host-staged, slow.  Numerically it matches RCCL for two ranks (one addition per element) and is
// deterministic for more.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

namespace host_transport {

constexpr int MAX_RANKS = 8;
constexpr size_t SLOT_BYTES = (size_t)4 << 20;   // per rank and chunk
struct Header {
    std::atomic<uint32_t> attached, count, gen;
    uint32_t nranks;
};
constexpr size_t SEG_BYTES = 4096 + MAX_RANKS * SLOT_BYTES;

struct Comm {
    Header* hdr = nullptr; uint8_t* slots = nullptr; int nranks = 1, rank = 0;
    void* pinned = nullptr;   // SLOT_BYTES of pinned staging for the result
};
enum { OK = 0, ERR_SYS = 1, ERR_TIMEOUT = 2, ERR_ARG = 3 };
inline const char* error_string(int e)
{
    switch (e) { case OK: return "ok"; case ERR_SYS: return "host transport: shared-memory segment could not be created / mapped";
                 case ERR_TIMEOUT: return "host transport: a rank did not reach the barrier within 120 s"; default: return "host transport: bad argument"; }
}

// the segment behind `name`: POSIX shared memory, or - where /dev/shm is missing or too small for it - a file under /tmp mapped MAP_SHARED
inline int seg_open(const char* name, bool create)
{
    int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
    if (fd >= 0 && create && ftruncate(fd, (off_t)SEG_BYTES) != 0) { close(fd); shm_unlink(name); fd = -1; }
    if (fd >= 0) return fd;
    char path[192];
    snprintf(path, sizeof path, "/tmp%s", name);
    fd = create ? open(path, O_CREAT | O_EXCL | O_RDWR, 0600) : open(path, O_RDWR);
    if (fd >= 0 && create && ftruncate(fd, (off_t)SEG_BYTES) != 0) { close(fd); unlink(path); fd = -1; }
    return fd;
}
inline void seg_unlink(const char* name)
{
    char path[192];
    snprintf(path, sizeof path, "/tmp%s", name);
    if (shm_unlink(name) != 0) (void)unlink(path);
}

inline int get_unique_id(rcclUniqueId* id)
{
    static std::atomic<unsigned> serial{0};
    memset(id->internal, 0, sizeof id->internal);
    const auto t = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/bdr_comm_%d_%u_%llx", (int)getpid(), serial.fetch_add(1), (unsigned long long)t);
    const int fd = seg_open(id->internal, true);   // (sparse: pages exist once they are touched)
    if (fd < 0) return ERR_SYS;
    close(fd);
    return OK;
}

inline int barrier(Comm* c)
{
    Header* h = c->hdr;
    const uint32_t gen = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.fetch_add(1, std::memory_order_release);
        return OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; h->gen.load(std::memory_order_acquire) == gen; ++spins) {
        if ((spins & 255u) == 255u) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ERR_TIMEOUT;
        }
    }
    return OK;
}

inline int comm_init_rank(rcclComm_t* out, int nranks, rcclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ERR_ARG;
    id.internal[sizeof id.internal - 1] = 0;
    const int fd = seg_open(id.internal, false);
    if (fd < 0) return ERR_SYS;
    void* m = mmap(nullptr, SEG_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ERR_SYS;
    Comm* c = new Comm();
    c->hdr = (Header*)m; c->slots = (uint8_t*)m + 4096; c->nranks = nranks; c->rank = rank;
    c->hdr->nranks = (uint32_t)nranks;
    if (hipHostMalloc(&c->pinned, SLOT_BYTES, hipHostMallocDefault) != hipSuccess) { munmap(m, SEG_BYTES); delete c; return ERR_SYS; }
    if (c->hdr->attached.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) seg_unlink(id.internal);   // every rank holds a mapping: the name can go
    *out = c;
    return barrier(c);   // like ncclCommInitRank: returns once every rank has joined
}

inline int comm_destroy(rcclComm_t cc)
{
    Comm* c = (Comm*)cc;
    if (!c) return OK;
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->hdr) munmap((void*)c->hdr, SEG_BYTES);
    delete c;
    return OK;
}

template <class T, class F>
inline int all_reduce_t(const void* send, void* recv, size_t count, Comm* c, hipStream_t st, F op)
{
    if (hipStreamSynchronize(st) != hipSuccess) return ERR_SYS;
    const size_t per = SLOT_BYTES / sizeof(T);
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, (const T*)send + off, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return ERR_SYS;
        if (int e = barrier(c)) return e;
        T* res = (T*)c->pinned;
        const T* s0 = (const T*)c->slots;
        for (size_t i = 0; i < n; ++i) res[i] = s0[i];
        for (int r = 1; r < c->nranks; ++r) {
            const T* sr = (const T*)(c->slots + (size_t)r * SLOT_BYTES);
            for (size_t i = 0; i < n; ++i) res[i] = op(res[i], sr[i]);
        }
        if (int e = barrier(c)) return e;   // every rank has read the slots: the next chunk may overwrite them
        if (hipMemcpyAsync((T*)recv + off, res, n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) return ERR_SYS;
        if (hipStreamSynchronize(st) != hipSuccess) return ERR_SYS;
    }
    return OK;
}

// ncclDataType_t: ncclInt32 = 2, ncclFloat32 = 7; ncclRedOp_t: ncclSum = 0, ncclMin = 3
inline int all_reduce(const void* send, void* recv, size_t count, int dtype, int op, rcclComm_t cc, hipStream_t st)
{
    Comm* c = (Comm*)cc;
    if (dtype == 7 && op == 0) return all_reduce_t<float>(send, recv, count, c, st, [](float a, float b) { return a + b; });
    if (dtype == 2 && op == 3) return all_reduce_t<int32_t>(send, recv, count, c, st, [](int32_t a, int32_t b) { return a < b ? a : b; });
    return ERR_ARG;
}

inline int broadcast(const void* send, void* recv, size_t count, int dtype, int root, rcclComm_t cc, hipStream_t st)
{
    Comm* c = (Comm*)cc;
    if (dtype != 7 || root < 0 || root >= c->nranks) return ERR_ARG;
    if (hipStreamSynchronize(st) != hipSuccess) return ERR_SYS;
    const size_t per = SLOT_BYTES / sizeof(float);
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (c->rank == root && hipMemcpy(c->slots, (const float*)send + off, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return ERR_SYS;
        if (int e = barrier(c)) return e;
        memcpy(c->pinned, c->slots, n * 4);
        if (int e = barrier(c)) return e;
        if (hipMemcpyAsync((float*)recv + off, c->pinned, n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ERR_SYS;
        if (hipStreamSynchronize(st) != hipSuccess) return ERR_SYS;
    }
    return OK;
}

}  // namespace host_transport
