// Multi-GPU parameter exchange over RCCL/xGMI: one process per GPU, the flat f32 parameter arena
// is all-reduced (north_star: periodic averaging) or broadcast from a root (the faithful
// learner->actors sync of border-async-trainer/src/async_trainer/base.rs:268-272, which ships
// NamedTensors over a crossbeam channel).  librccl is resolved lazily with dlopen so that
// single-GPU users never load it, and so that inside a PyTorch process the already-loaded
// librccl.so.1 (same soname) is reused.
#include <dlfcn.h>

#include "common.hpp"

using namespace bdr;

namespace bdr {
float* agent_arena(bdr_agent* a, int which, size_t* n_floats, hipStream_t* stream, int* device);
int32_t agent_scale(bdr_agent* a, float* p, size_t n, float s);
struct XSeg { size_t off, n; };
int agent_exchange_plan(bdr_agent* a, int which, XSeg* segs, int cap, hipStream_t* comm);
int32_t agent_exchange_begin(bdr_agent* a, int seg);
int32_t agent_exchange_end(bdr_agent* a, int seg);
int32_t scale_on(hipStream_t st, float* p, size_t n, float s);
void agent_set_grad_comm(bdr_agent* a, void* comm, int32_t (*reduce)(bdr_agent*, void*));
}

namespace {
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
// ncclDataType_t: ncclFloat32 = 7; ncclRedOp_t: ncclSum = 0
constexpr int kNcclFloat = 7, kNcclInt32 = 2, kNcclSum = 0, kNcclMin = 3;

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(rcclUniqueId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
}  // namespace
#ifdef BDR_COMM_HOST_TRANSPORT
#include "comm_host_transport.hpp"   // (test build only: see the file's header)
#endif
namespace {

int32_t load_rccl()
{
    if (g_rccl.h) return BDR_OK;
#ifdef BDR_COMM_HOST_TRANSPORT
    {   // the test library: the same six entry points over host shared memory, so that N ranks may share one GPU
        Rccl r; r.h = (void*)&g_rccl;
        r.GetUniqueId = host_transport::get_unique_id; r.CommInitRank = host_transport::comm_init_rank; r.CommDestroy = host_transport::comm_destroy;
        r.AllReduce = host_transport::all_reduce; r.Broadcast = host_transport::broadcast; r.GetErrorString = host_transport::error_string;
        g_rccl = r;
        fprintf(stderr, "border_amd: this is the HOST-TRANSPORT TEST BUILD of the communicator (libborder_amd_hostcomm.so): collectives are staged through host shared memory\n");
        return BDR_OK;
    }
#endif
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(BDR_ERR_COMM, "cannot load librccl: %s", dlerror());
    Rccl r; r.h = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) return fail(BDR_ERR_COMM, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce"); SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = r;
    return BDR_OK;
}
#define BDR_NCCL(expr)                                                                                             \
    do {                                                                                                           \
        int e__ = (expr);                                                                                          \
        if (e__ != 0) return fail(BDR_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(e__));               \
    } while (0)
}  // namespace

// Hardware queues for the N>1 path.  The overlapped parameter exchange adds a communication queue to the agent's three
// streams and the replay buffer's: five streams that must not share an in-order HSA queue, and HIP multiplexes streams onto
// GPU_MAX_HW_QUEUES (default 4) queues per process.  The variable is read when the HIP runtime initialises (its first API
// call) and it is process-wide - it changes the queue allocation of every HIP user in the process, torch under torchrun
// included - so the library only touches it on an EXPLICIT request of the host: BDR_REQUEST_HW_QUEUES=<n> (or BDR_NRANKS > 1,
// the library's own variable) in the environment when the library is loaded, before the process's first HIP call.  It never
// infers the request from a launcher's WORLD_SIZE / PMI_SIZE.  Hosts normally export GPU_MAX_HW_QUEUES=8 themselves (bench.py,
// tests/test_gpu_multi.py); without enough queues the agent detects the aliasing and falls back to the
// in-stream exchange, and bdr_comm_init_rank says so.
namespace {
int env_int(const char* k) { const char* e = getenv(k); return e ? atoi(e) : 0; }
__attribute__((constructor)) void bdr_request_hw_queues()
{
    const int want = env_int("BDR_REQUEST_HW_QUEUES") > 0 ? env_int("BDR_REQUEST_HW_QUEUES") : (env_int("BDR_NRANKS") > 1 ? 8 : 0);
    if (want > 0) { char v[16]; snprintf(v, sizeof v, "%d", want); setenv("GPU_MAX_HW_QUEUES", v, /*overwrite=*/0); }
}
}  // namespace

struct bdr_comm {
    rcclComm_t comm = nullptr; int nranks = 1, rank = 0, device = 0;
    int32_t* d_flag = nullptr; hipStream_t flag_stream = nullptr;   // bdr_comm_agree
};

extern "C" {

int32_t bdr_comm_get_unique_id(uint8_t id[BDR_UNIQUE_ID_BYTES])
{
    BDR_REQUIRE(id, "null argument");
    BDR_TRY(load_rccl());
    rcclUniqueId u;
    BDR_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, BDR_UNIQUE_ID_BYTES);
    return BDR_OK;
}

int32_t bdr_comm_init_rank(const uint8_t id[BDR_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank, int32_t device, bdr_comm** out)
{
    BDR_REQUIRE(id && out, "null argument");
    BDR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    BDR_TRY(ensure_device(device));
    BDR_TRY(load_rccl());
    rcclUniqueId u;
    memcpy(u.internal, id, BDR_UNIQUE_ID_BYTES);
    bdr_comm* c = new bdr_comm();
    c->nranks = nranks; c->rank = rank; c->device = device;
    int e = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    if (e != 0) { delete c; return fail(BDR_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(e)); }
    if (nranks > 1) {
        static bool noted = false;
        const int q = env_int("GPU_MAX_HW_QUEUES");
        if (!noted && q < 5) {
            noted = true;
            fprintf(stderr, "border_amd: GPU_MAX_HW_QUEUES=%s with %d ranks: the overlapped parameter exchange needs 5 hardware queues; "
                            "export GPU_MAX_HW_QUEUES=8 before the process's first HIP call (or BDR_REQUEST_HW_QUEUES=8 before the library is "
                            "loaded); agents fall back to the in-stream exchange\n",
                    getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset (4)", nranks);
        }
    }
    *out = c;
    return BDR_OK;
}

int32_t bdr_comm_destroy(bdr_comm* c)
{
    if (!c) return BDR_OK;
    if (c->flag_stream) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->flag_stream); (void)hipStreamDestroy(c->flag_stream); }
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return BDR_OK;
}

// Agreement before a collective (bdr_learner_ops::agree, ParamExchange.rccl_or_raise's counterpart on the data plane): MIN
// all-reduce of one int32 on the communicator's own small stream, read back synchronously.  Every rank calls it the same number
// of times; a rank that failed passes 0 and every rank learns it.
int32_t bdr_comm_agree(bdr_comm* c, int32_t local_ok, int32_t* all_ok)
{
    BDR_REQUIRE(c && all_ok, "null argument");
    BDR_HIP(hipSetDevice(c->device));
    if (!c->d_flag) {
        BDR_HIP(hipMalloc((void**)&c->d_flag, sizeof(int32_t)));
        BDR_HIP(hipStreamCreateWithFlags(&c->flag_stream, hipStreamNonBlocking));
    }
    const int32_t v = local_ok ? 1 : 0;
    BDR_HIP(hipMemcpyAsync(c->d_flag, &v, sizeof v, hipMemcpyHostToDevice, c->flag_stream));
    BDR_NCCL(g_rccl.AllReduce(c->d_flag, c->d_flag, 1, kNcclInt32, kNcclMin, c->comm, c->flag_stream));
    int32_t r = 0;
    BDR_HIP(hipMemcpyAsync(&r, c->d_flag, sizeof r, hipMemcpyDeviceToHost, c->flag_stream));
    BDR_HIP(hipStreamSynchronize(c->flag_stream));
    *all_ok = r;
    return BDR_OK;
}

int32_t bdr_agent_allreduce_params(bdr_agent* a, bdr_comm* c, int32_t which)
{
    BDR_REQUIRE(a && c, "null argument");
    size_t n = 0; hipStream_t s = nullptr; int dev = 0;
    float* p = agent_arena(a, which, &n, &s, &dev);
    BDR_REQUIRE(p, "unknown arena");
    BDR_REQUIRE(dev == c->device, "agent and communicator live on different devices");
    BDR_HIP(hipSetDevice(dev));
    // Overlapped form: the agent names segments whose exchange may start before its step has finished (the l1 / l2 parameters
    // are final once their Adam pass on the weight-gradient queue is done, well before the conv dX chain ends) and may finish
    // after the next step has started (the next forward waits per segment): one collective per segment on the agent's
    // communication queue, bracketed by the agent's own ordering hooks.
    XSeg segs[4]; hipStream_t cs = nullptr;
    const int ns = agent_exchange_plan(a, which, segs, 4, &cs);
    if (ns > 0) {
        for (int k = 0; k < ns; ++k) {
            BDR_TRY(agent_exchange_begin(a, k));
            BDR_NCCL(g_rccl.AllReduce(p + segs[k].off, p + segs[k].off, segs[k].n, kNcclFloat, kNcclSum, c->comm, cs));
            BDR_TRY(scale_on(cs, p + segs[k].off, segs[k].n, 1.0f / (float)c->nranks));
            BDR_TRY(agent_exchange_end(a, k));
        }
        return BDR_OK;
    }
    BDR_NCCL(g_rccl.AllReduce(p, p, n, kNcclFloat, kNcclSum, c->comm, s));
    return agent_scale(a, p, n, 1.0f / (float)c->nranks);
}

// Synchronous data-parallel mode: every Agent::opt of `a` all-reduces its gradient arena (sum, 1/nranks) between backward and the
// optimizer step (agent_base.hpp grad_comm).  c == NULL switches back to independent steps.
static int32_t grad_reduce_rccl(bdr_agent* a, void* comm)
{
    bdr_comm* c = (bdr_comm*)comm;
    size_t n = 0; hipStream_t s = nullptr; int dev = 0;
    float* g = agent_arena(a, 4, &n, &s, &dev);
    BDR_REQUIRE(g, "agent has no gradient arena");
    BDR_NCCL(g_rccl.AllReduce(g, g, n, kNcclFloat, kNcclSum, c->comm, s));
    if (c->nranks == 1) return BDR_OK;
    return agent_scale(a, g, n, 1.0f / (float)c->nranks);
}

int32_t bdr_agent_set_grad_comm(bdr_agent* a, bdr_comm* c)
{
    BDR_REQUIRE(a, "null agent");
    if (c) {
        int dev = 0;
        BDR_REQUIRE(agent_arena(a, 4, nullptr, nullptr, &dev), "agent has no gradient arena");
        BDR_REQUIRE(dev == c->device, "agent and communicator live on different devices");
    }
    agent_set_grad_comm(a, c, c ? grad_reduce_rccl : nullptr);
    return BDR_OK;
}

int32_t bdr_agent_broadcast_params(bdr_agent* a, bdr_comm* c, int32_t which, int32_t root)
{
    BDR_REQUIRE(a && c, "null argument");
    BDR_REQUIRE(root >= 0 && root < c->nranks, "bad root");
    size_t n = 0; hipStream_t s = nullptr; int dev = 0;
    float* p = agent_arena(a, which, &n, &s, &dev);
    BDR_REQUIRE(p, "unknown arena");
    BDR_HIP(hipSetDevice(dev));
    BDR_NCCL(g_rccl.Broadcast(p, p, n, kNcclFloat, root, c->comm, s));
    return BDR_OK;
}

}  // extern "C"
