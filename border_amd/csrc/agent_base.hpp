// Common agent plumbing of libborder_amd.so: the polymorphic handle behind `bdr_agent*`, per-kernel
// HIP-event profiling brackets, launch macros, and the optimizer kernels every agent shares.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "chacha.hpp"
#include "common.hpp"
#include "igemm.hpp"
#include "step_graph.hpp"

// DqnExplorer / IqnExplorer state (dqn/explorer.rs:8-120, iqn/explorer.rs:9-108).  The reference draws from
// fastrand's global, unseeded generator; here the draws come from a seeded ChaCha12 stream (same generator
// as the replay buffer's StdRng), so a run is reproducible and testable draw by draw.
struct Explorer {
    int32_t kind = BDR_EXPLORER_SOFTMAX;      // dqn/config.rs:93 default explorer
    double eps_start = 1.0, eps_final = 0.02; // explorer.rs:47-50
    uint64_t final_step = 100000;
    uint64_t n_calls = 0;                     // the reference's `n_opts` field: counts action() calls
    bdr::ChaChaKey key{};
    uint64_t word_pos = 0;
    uint32_t next_u32() { return bdr::chacha12_word(key, word_pos++); }
    uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | ((uint64_t)next_u32() << 32); }   // rand: low word first
    double f64() { return (double)(next_u64() >> 12) * (1.0 / 4503599627370496.0); }     // 52 random bits, [0,1)
    float f32() { return (float)(next_u32() >> 9) * (1.0f / 8388608.0f); }               // 23 random bits
    // unbiased integer in [0,n): Lemire's multiply-shift with rejection (what fastrand::u32(..n) does)
    uint32_t below(uint32_t n)
    {
        uint64_t m = (uint64_t)next_u32() * n;
        uint32_t lo = (uint32_t)m;
        if (lo < n) {
            const uint32_t t = (0u - n) % n;
            while (lo < t) { m = (uint64_t)next_u32() * n; lo = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
};

struct ProfSlot { std::string name; hipEvent_t e0, e1; double ms = 0; uint64_t count = 0; };

// The opaque `bdr_agent` of include/border_amd.h.  Concrete agents: DqnCnn (dqn.hip), DqnMlp / Sac
// (mlp_agents.hip), Iqn (iqn.hip).
// One observation row of a compiled loop (trainer.hip, async_trainer.hip): host bytes, or - for environments with device-resident
// observations (bdr_env_vtable::obs_on_device) - a device buffer of the same size.
struct ObsRow {
    bool dev = false; size_t bytes = 0; int device = 0;
    std::vector<uint8_t> h; uint8_t* d = nullptr;
    ObsRow() = default;
    ObsRow(const ObsRow&) = delete; ObsRow& operator=(const ObsRow&) = delete;
    ~ObsRow() { if (d) { (void)hipSetDevice(device); (void)hipFree(d); } }
    int32_t init(bool on_device, int dev_ix, size_t n)
    {
        dev = on_device; device = dev_ix; bytes = n;
        if (!dev) { h.assign(n, 0); return BDR_OK; }
        BDR_HIP(hipSetDevice(device));
        BDR_HIP(hipMalloc((void**)&d, n));
        return BDR_OK;
    }
    void* p() { return dev ? (void*)d : (void*)h.data(); }
    const void* p() const { return dev ? (const void*)d : (const void*)h.data(); }
    void swap(ObsRow& o) { std::swap(h, o.h); std::swap(d, o.d); }   // `a = b` of the reference where b is dead afterwards
};

// rows -> pinned host memory, then the sequence number the host waits for (system-scope release: the rows are visible before it); the
// device-side error words ride along (bdr_agent::err_poll reads them)
constexpr size_t ROWS_PINNED_MAX_FLOATS = 4096;
// the device's error words -> their pinned host mirror (bdr_agent::err_poll, every ERR_POLL_INTERVAL opts): a one-wave kernel in the
// agent's stream.  (A device -> host copy COMMAND there made the queue wait for it: 0.75 ms under the kernel tracer, the 0.8 ms gate
// launches of the default-schedule trace - the other queue waiting through it; profiles/gate_max_r05.md.)
static __global__ __launch_bounds__(64) void k_err_mirror(const unsigned* dev_err, unsigned* host_err, int n)
{
    if ((int)threadIdx.x < n) host_err[threadIdx.x] = dev_err[threadIdx.x];
}

static __global__ __launch_bounds__(256) void k_publish_rows(const float* src, float* dst_host, unsigned n, unsigned* seq_host, unsigned seq, const unsigned* dev_err, int n_err)
{
    for (unsigned i = threadIdx.x; i < n; i += 256) dst_host[i] = src[i];
    if (dev_err && (int)threadIdx.x < n_err) seq_host[4 + threadIdx.x] = dev_err[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct bdr_agent {
    int32_t device = 0;
    hipStream_t stream = nullptr;
    bool train = false;
    uint64_t n_opts = 0;
    int32_t ckpt_format = BDR_CKPT_TCH;
    // Policy::sample state (dqn/base.rs:211-242)
    Explorer explorer;
    uint64_t n_samples_act = 0, n_samples_best_act = 0;
    void* act_stage = nullptr;      // device staging for the observations of Policy::sample (grown on demand)
    size_t act_stage_bytes = 0;
    // prioritized replay (dqn/base.rs:123-145): |pred - tgt| of the last update, and staged host weights
    float* td_abs = nullptr; float* w_stage = nullptr; size_t td_cap = 0;
    int32_t td_buffer(size_t n)
    {
        if (n > td_cap) {
            BDR_HIP(hipStreamSynchronize(stream));
            (void)hipFree(td_abs); (void)hipFree(w_stage); td_abs = w_stage = nullptr; td_cap = 0;
            const size_t cap = std::max(n, (size_t)256);
            BDR_HIP(hipMalloc((void**)&td_abs, cap * 4));
            BDR_HIP(hipMalloc((void**)&w_stage, cap * 4));
            td_cap = cap;
        }
        return BDR_OK;
    }
    // profiling
    bool prof = false;
    std::vector<ProfSlot> slots;
    size_t slot_cursor = 0;
    // Device-side error words (kernels never abort; they flag and keep memory safe):
    //   [ERR_ACTION]  an action index outside [0, n_actions) reached a TD kernel (the reference's gather would raise);
    //                 the kernels clamp it so that no out-of-bounds access happens
    //   [ERR_GATE]    1 + flag id of a cross-queue gate that timed out (DqnCnn schedule 3): parameter updates are skipped
    //                 from then on until the host has seen the error
    //   [ERR_NONFINITE] reserved
    // Checked after every host synchronisation (err_check) and, without synchronising, every ERR_POLL_INTERVAL opts through
    // an asynchronous copy into pinned memory (err_poll) - so loops that never synchronise (Agent::opt only) still fail
    // within a few hundred steps instead of training on garbage.
    enum { ERR_ACTION = 0, ERR_GATE = 1, ERR_NONFINITE = 2, ERR_WORDS = 4, ERR_POLL_INTERVAL = 256 };
    unsigned* dev_err = nullptr;        // [ERR_WORDS] device
    unsigned* host_err = nullptr;       // [ERR_WORDS] pinned mirror (last asynchronous read-back)
    unsigned* host_err_dev = nullptr;   // the mirror's device address (k_err_mirror)
    uint64_t err_poll_count = 0;
    bool rec_opt = false;               // record() is being called by Agent::opt_with_record (dqn/base.rs:316-342)
    uint64_t last_replay_uid = 0;       // uid of the buffer of the last opt (its PER error flag is checked with the agent's); a uid, not a
                                        // pointer: the buffer may be destroyed before the agent's next synchronising call (replay_lookup)

    virtual ~bdr_agent()
    {
        if (act_stage) (void)hipFree(act_stage);
        (void)hipFree(td_abs); (void)hipFree(w_stage);
        (void)hipFree(dev_err);
        if (host_err) (void)hipHostFree(host_err);
        if (rows_host) (void)hipHostFree(rows_host);
        if (act_pin) (void)hipHostFree(act_pin);
    }
    int32_t err_init()
    {
        BDR_HIP(hipMalloc((void**)&dev_err, ERR_WORDS * sizeof(unsigned)));
        BDR_HIP(hipMemset(dev_err, 0, ERR_WORDS * sizeof(unsigned)));
        BDR_HIP(hipHostMalloc((void**)&host_err, ERR_WORDS * sizeof(unsigned), hipHostMallocDefault));
        memset(host_err, 0, ERR_WORDS * sizeof(unsigned));
        return BDR_OK;
    }
    // turns the error words `w` into a status; clears them on the device when something was set
    int32_t err_report(const unsigned* w);
    int32_t err_check();   // the stream has just been synchronised: read, report, clear
    int32_t err_poll();    // no synchronisation: look at the last asynchronous read-back, enqueue the next one when due
    // Result rows of an ACTING call (Policy::sample: a handful of floats once per environment step) -> host.  Up to ROWS_PINNED_MAX_FLOATS
    // go through pinned host memory: a one-workgroup kernel behind the forward pass stores the rows, the error words and then a sequence
    // number there, and this thread waits for the number - instead of a copy command plus a stream synchronisation (~25 us of host
    // latency for 24 bytes).  Larger results take the copy.
    float* rows_host = nullptr; float* rows_dev = nullptr; unsigned rows_seq = 0;   // [0] sequence word, [4...] error words, [16...] rows
    int32_t rows_to_host(const float* dev_rows, float* out, size_t n_floats)
    {
        if (n_floats > ROWS_PINNED_MAX_FLOATS) {
            BDR_HIP(hipMemcpyAsync(out, dev_rows, n_floats * 4, hipMemcpyDeviceToHost, stream));
            BDR_HIP(hipStreamSynchronize(stream));
            return BDR_OK;
        }
        if (!rows_host) {
            BDR_HIP(hipHostMalloc((void**)&rows_host, (ROWS_PINNED_MAX_FLOATS + 16) * sizeof(float), hipHostMallocMapped));
            memset(rows_host, 0, (ROWS_PINNED_MAX_FLOATS + 16) * sizeof(float));
            BDR_HIP(hipHostGetDevicePointer((void**)&rows_dev, rows_host, 0));
        }
        const unsigned seq = ++rows_seq;
        hipLaunchKernelGGL(k_publish_rows, dim3(1), dim3(256), 0, stream, dev_rows, rows_dev + 16, (unsigned)n_floats, reinterpret_cast<unsigned*>(rows_dev), seq,
                           (const unsigned*)dev_err, (int)ERR_WORDS);
        BDR_HIP(hipGetLastError());
        return rows_wait(seq, out, n_floats);
    }
    // wait for sequence number `seq` in the pinned area, then copy the rows (and this call's view of the error words) out
    int32_t rows_wait(unsigned seq, float* out, size_t n_floats)
    {
        const volatile unsigned* done = reinterpret_cast<const volatile unsigned*>(rows_host);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; (int)(*done - seq) < 0; ++spins) {
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {   // a lost kernel: let the stream report
                BDR_HIP(hipStreamSynchronize(stream));
                if ((int)(*done - seq) < 0) return ::bdr::fail(BDR_ERR_HIP, "the result rows of an acting call never arrived in host memory");
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(out, rows_host + 16, n_floats * sizeof(float));
        if (dev_err && host_err) {   // this call's view of the error words: the caller polls them instead of copying them once more
            for (int i = 0; i < ERR_WORDS; ++i) reinterpret_cast<volatile unsigned*>(host_err)[i] = reinterpret_cast<const volatile unsigned*>(rows_host)[4 + i];
            err_fresh = true;
        }
        return BDR_OK;
    }
    // Host rows of an acting call -> pinned memory the device reads in place (a host -> device copy command from pageable memory costs
    // ~10-20 us of host time for a few rows; the first kernel of the forward fetches them over PCIe instead).  Acting calls are synchronous -
    // the caller waits for their result - so the area is free again whenever the next call fills it.
    static constexpr size_t HOST_ROWS_PINNED_MAX = 256 * 1024;
    uint8_t* act_pin = nullptr; uint8_t* act_pin_dev = nullptr; size_t act_pin_bytes = 0;
    int32_t host_rows_pinned(const void* rows, size_t bytes, const uint8_t** dev)
    {
        if (bytes > act_pin_bytes) {
            BDR_HIP(hipStreamSynchronize(stream));
            if (act_pin) (void)hipHostFree(act_pin);
            act_pin = nullptr; act_pin_dev = nullptr; act_pin_bytes = 0;
            const size_t cap = std::max(bytes, (size_t)64 * 1024);
            BDR_HIP(hipHostMalloc((void**)&act_pin, cap, hipHostMallocMapped));
            BDR_HIP(hipHostGetDevicePointer((void**)&act_pin_dev, act_pin, 0));
            act_pin_bytes = cap;
        }
        memcpy(act_pin, rows, bytes);
        *dev = act_pin_dev;
        return BDR_OK;
    }
    bool err_fresh = false;   // host_err was refreshed by the call in progress (an acting call's Q rows brought the words along): poll, do not copy again
    virtual void on_gate_timeout() {}   // DqnCnn: fall back to event ordering
    // Called by err_report BEFORE it clears the error words: every queue of the agent must be idle by then.  While the words are up
    // the waits queued on the other queues return at once (poison); cleared under them they would start a fresh time limit and
    // could poison the agent a second time, and kernels still queued there would write the batch sets the fallback schedule reads.
    virtual void drain_queues() { (void)hipStreamSynchronize(stream); }
    // Synchronous data-parallel mode (SURVEY.md 8(e) "Collective", last sentence): opt() = backward -> all-reduce of the
    // gradient arena over the communicator (sum, then 1/N) -> optimizer step, so that N ranks with batch B/N each take exactly
    // the step one rank takes on the concatenated batch.  grad_reduce is installed by bdr_agent_set_grad_comm (comm.hip).
    void* grad_comm = nullptr;
    int32_t (*grad_reduce)(bdr_agent*, void*) = nullptr;
    // Overlapped parameter exchange (bdr_agent_allreduce_params on arena 0): an agent that can order a communication queue
    // against its compute queues returns the segments of the arena, each with hooks the collective is bracketed by -
    //   begin(seg): make `comm` wait until the segment's last writer / reader of this step is done
    //   end(seg):   publish "segment exchanged" so that the next step's first reader of the segment waits for it
    // - and the collective runs on `comm` beside the rest of the step.  n == 0: no plan, the exchange runs in-stream.
    struct ExchangeSeg { size_t off, n; };
    virtual int exchange_plan(int /*which*/, ExchangeSeg* /*segs*/, int /*cap*/, hipStream_t* /*comm*/) { return 0; }
    virtual int32_t exchange_begin(int /*seg*/) { return BDR_OK; }
    virtual int32_t exchange_end(int /*seg*/) { return BDR_OK; }
    // update_critic up to `loss.backward()` on a host minibatch: gradients land in the gradient arena, parameters and
    // optimizer state are untouched; apply_grads = the optimizer step on whatever the gradient arena holds + opt_ bookkeeping
    virtual int32_t grads_on_batch(uint64_t, const void*, const int64_t*, const void*, const float*, const int8_t*)
    {
        return ::bdr::fail(BDR_ERR_INVALID, "this agent kind has no split backward / optimizer step");
    }
    virtual int32_t apply_grads() { return ::bdr::fail(BDR_ERR_INVALID, "this agent kind has no split backward / optimizer step"); }
    int32_t act_buffer(size_t bytes, void** out)
    {
        if (bytes > act_stage_bytes) {
            if (act_stage) { BDR_HIP(hipStreamSynchronize(stream)); BDR_HIP(hipFree(act_stage)); act_stage = nullptr; act_stage_bytes = 0; }
            const size_t cap = std::max(bytes, (size_t)1 << 16);
            BDR_HIP(hipMalloc(&act_stage, cap));
            act_stage_bytes = cap;
        }
        *out = act_stage;
        return BDR_OK;
    }
    // Observation rows of an acting call -> dst (device, contiguous).  Host rows by default; inside a *_device entry point
    // (bdr_agent_sample_device ...) the rows already live in HBM, obs_row_stride bytes apart, and never touch the host.
    bool obs_rows_on_device = false;
    uint64_t obs_row_stride = 0;
    // scope of a *_device entry point: the flag is down again on every way out of the nested host-row call
    struct DeviceRowsScope {
        bdr_agent* a;
        DeviceRowsScope(bdr_agent* a_, uint64_t stride) : a(a_) { a->obs_rows_on_device = true; a->obs_row_stride = stride; }
        ~DeviceRowsScope() { a->obs_rows_on_device = false; a->obs_row_stride = 0; }
        DeviceRowsScope(const DeviceRowsScope&) = delete; DeviceRowsScope& operator=(const DeviceRowsScope&) = delete;
    };
    // obs_dev must be device memory of the agent's GPU, rows a multiple of 4 bytes apart
    int32_t check_device_rows(const void* obs_dev, uint64_t row_stride) const
    {
        if (!obs_dev) return ::bdr::fail(BDR_ERR_INVALID, "null argument");
        if (row_stride == 0 || row_stride % 4 != 0) return ::bdr::fail(BDR_ERR_INVALID, "row_stride must be a positive multiple of 4 bytes");
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, obs_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != device)
            return ::bdr::fail(BDR_ERR_INVALID, "obs_dev is not device memory of the agent's GPU (host rows go through the host-row entry point)");
        return BDR_OK;
    }
    int32_t stage_obs(void* dst, const void* src, size_t row_bytes, uint64_t n, hipStream_t st)
    {
        if (obs_rows_on_device && obs_row_stride < row_bytes)
            return ::bdr::fail(BDR_ERR_INVALID, "row_stride (%llu bytes) is smaller than the agent's observation row (%llu bytes)",
                               (unsigned long long)obs_row_stride, (unsigned long long)row_bytes);
        if (!obs_rows_on_device) BDR_HIP(hipMemcpyAsync(dst, src, n * row_bytes, hipMemcpyHostToDevice, st));
        else if (obs_row_stride == row_bytes) BDR_HIP(hipMemcpyAsync(dst, src, n * row_bytes, hipMemcpyDeviceToDevice, st));
        else BDR_HIP(hipMemcpy2DAsync(dst, row_bytes, src, obs_row_stride, row_bytes, n, hipMemcpyDeviceToDevice, st));
        return BDR_OK;
    }
    // the rows can be read in place (device rows that are already contiguous)
    bool obs_in_place(size_t row_bytes) const { return obs_rows_on_device && obs_row_stride == row_bytes; }
    virtual const char* kind() const = 0;
    virtual int32_t opt(bdr_replay* r) = 0;                       // Agent::opt, asynchronous
    virtual int32_t after_sync() { return BDR_OK; }               // device-side error flags, checked by bdr_agent_sync
    virtual int32_t record(float* out, int cap, int* n) = 0;      // scalars of the last update (syncs)
    virtual void record_keys(std::vector<std::string>& keys) = 0; // names of those scalars, in order
    // n draws of the agent's own device noise stream (SAC: N(0,1) of action_logp; IQN: U[0,1) percent points), advancing it
    virtual int32_t noise(float*, size_t) { return ::bdr::fail(BDR_ERR_INVALID, "this agent draws no device noise"); }
    virtual uint64_t param_count(int which) = 0;                  // reference-layout element count
    virtual int32_t get_params(int which, float* out, uint64_t n) = 0;
    virtual int32_t set_params(int which, const float* in, uint64_t n) = 0;
    virtual float* arena(int which, size_t* n) = 0;               // flat device arena (kernel layout)
    virtual void arena_released(int) {}                           // ... and the caller handed it back (bdr_agent_arena_release)
    virtual void arena_escaped(int) {}                            // the raw pointer of arena `which` left the library (bdr_agent_arena_device_ptr)
    virtual int32_t save(const char* dir) = 0;
    virtual int32_t load(const char* dir) = 0;
};

namespace bdr {

// TD loss of one row (dqn/base.rs:123-151).  Without importance weights: smooth_l1 / mse of (pred, tgt).
// With weights (PER): td = |pred - tgt| (clipped to [cmin, cmax] when clip_td_err is set), x = w * td,
// loss = smooth_l1(x, 0) / mse(x, 0); the gradient follows autograd through abs / clip (clip passes the
// gradient inside its closed range).  Returns dLoss_row/dpred (before the 1/B of Reduction::Mean).
struct TdLossIn { int loss_kind; int weighted; float w; int has_clip; float cmin, cmax; };
// r + (1 - is_terminated) * gamma * q'   (dqn/base.rs:104).  One definition for every kernel that forms it, without fused
// multiply-add contraction: hipcc contracts across statements, and the same expression compiled into two kernels may round
// differently - the paths that must agree bit for bit (one-workgroup step, layer-by-layer, row-block head) all call this.
__device__ __forceinline__ float td_target(float reward, float not_done, float gamma, float qn)
{
#pragma clang fp contract(off)
    const float k = not_done * gamma;
    const float c = k * qn;
    return reward + c;
}
__device__ __forceinline__ float td_loss_row(float pred, float tgt, const TdLossIn& c, float& lossb, float& td_abs)
{
#pragma clang fp contract(off)
    const float d = pred - tgt;
    if (!c.weighted) {
        td_abs = fabsf(d);
        if (c.loss_kind == 1) {   // smooth_l1_loss(beta=1.0)
            const float z = fabsf(d);
            lossb = z < 1.f ? 0.5f * z * z : z - 0.5f;
            return z < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        }
        lossb = d * d;            // mse_loss
        return 2.f * d;
    }
    const float raw = fabsf(d);
    float td = raw, pass = 1.f;
    if (c.has_clip) { td = fminf(fmaxf(raw, c.cmin), c.cmax); pass = (raw >= c.cmin && raw <= c.cmax) ? 1.f : 0.f; }
    td_abs = td;
    const float x = c.w * td;
    float dx;
    if (c.loss_kind == 1) {
        const float z = fabsf(x);
        lossb = z < 1.f ? 0.5f * z * z : z - 0.5f;
        dx = z < 1.f ? x : (x > 0.f ? 1.f : -1.f);
    } else {
        lossb = x * x;
        dx = 2.f * x;
    }
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return dx * c.w * pass * sgn;
}

struct Bracket {
    bdr_agent* a; ProfSlot* s = nullptr;
    Bracket(bdr_agent* ag, const char* name) : a(ag)
    {
        if (!a->prof) return;
        if (a->slot_cursor >= a->slots.size()) {
            ProfSlot ns; ns.name = name;
            (void)hipEventCreate(&ns.e0); (void)hipEventCreate(&ns.e1);
            a->slots.push_back(ns);
        }
        s = &a->slots[a->slot_cursor++];
        (void)hipEventRecord(s->e0, a->stream);
    }
    ~Bracket()
    {
        if (!s) return;
        (void)hipEventRecord(s->e1, a->stream);
    }
};

inline void prof_collect(bdr_agent* a)
{
    if (!a->prof) return;
    (void)hipStreamSynchronize(a->stream);
    for (size_t i = 0; i < a->slot_cursor; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, a->slots[i].e0, a->slots[i].e1) == hipSuccess) { a->slots[i].ms += ms; a->slots[i].count++; }
    }
    a->slot_cursor = 0;
}

inline int32_t alloc_f(float** p, size_t n)
{
    BDR_HIP(hipMalloc((void**)p, (n < 4 ? 4 : n) * sizeof(float)));
    return BDR_OK;
}

// named f32 tensor container used by every agent's save/load (reference variable names)
struct NamedTensor { std::string name; std::vector<uint64_t> dims; };
// util.rs:64-80 param_stats: `<var>_mean` = v.mean(), `<var>_std` = v.std(false) (population) of every variable, appended to
// `out` in the variables' reference order (data: the reference-layout parameter vector described by meta)
inline void param_stats(const std::vector<NamedTensor>& meta, const float* data, std::vector<float>& out)
{
    size_t o = 0;
    for (const auto& m : meta) {
        size_t n = 1;
        for (auto d : m.dims) n *= d;
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += data[o + i];
        const double mean = n ? s / (double)n : 0.0;
        double q = 0.0;
        for (size_t i = 0; i < n; ++i) { const double d = data[o + i] - mean; q += d * d; }
        out.push_back((float)mean);
        out.push_back((float)std::sqrt(n ? q / (double)n : 0.0));
        o += n;
    }
}
inline void param_stat_keys(const std::vector<NamedTensor>& meta, std::vector<std::string>& keys)
{
    for (const auto& m : meta) { keys.push_back(m.name + "_mean"); keys.push_back(m.name + "_std"); }
}
int32_t save_named(const std::string& path, const std::vector<NamedTensor>& meta, const float* data, size_t n);
int32_t load_named(const std::string& path, const std::vector<NamedTensor>& meta, float* data, size_t n);
// "<dir>/<stem>.pt.tch" (the reference's names) or "<dir>/<stem>.safetensors" by bdr_agent_set_checkpoint_format; the load
// path falls back to the other container when only that one exists
std::string ckpt_save_path(const bdr_agent* a, const char* dir, const std::string& stem);
std::string ckpt_load_path(const bdr_agent* a, const char* dir, const std::string& stem);

}  // namespace bdr

#define LAUNCH_ON(st, kernel, grid, args)                                                           \
    do {                                                                                           \
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, args);                                   \
        BDR_HIP(hipGetLastError());                                                                \
    } while (0)
#define LAUNCH(kernel, grid, args) LAUNCH_ON(a->stream, kernel, grid, args)
// flags: 0 or hipExtAnyOrderLaunch, stop: event completed by the kernel's own packet or nullptr (see launch_igemm)
#define LAUNCH_FL(st, flags, stop, kernel, grid, block, ...)                                          \
    do {                                                                                           \
        hipExtLaunchKernelGGL(kernel, grid, block, 0, st, nullptr, stop, flags, __VA_ARGS__);         \
        BDR_HIP(hipGetLastError());                                                                \
    } while (0)

namespace {
using namespace bdr;

// ================================================================================================
// Adam (opt.rs:35 -> libtorch Adam::step) and track (util.rs:31-45) over the flat arena
// ================================================================================================
struct AdamScalars { float b1, omb1, b2, omb2, sqrt_bc2, eps, neg_step, wd_mul; };

// libtorch Adam::step for one element.  Every operation rounds on its own (no fused multiply-add contraction): the fused and
// the split optimizer paths (k_reduce_adam / k_adam, synchronous-DP mode) must agree bit for bit whichever kernel an element is
// updated by, and separately rounded f32 operations are also what the reference's vectorised CPU kernels compute.
__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, const AdamScalars& s)
{
#pragma clang fp contract(off)
    p = p * s.wd_mul;                             // AdamW decoupled decay (1 for Adam)
    const float mg = g * s.omb1;
    m = m * s.b1 + mg;                            // exp_avg.mul_(b1).add_(g, 1-b1)
    const float gg = s.omb2 * g * g;
    v = v * s.b2 + gg;                            // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = __fsqrt_rn(v) / s.sqrt_bc2 + s.eps;
    const float upd = s.neg_step * m / denom;
    p = p + upd;                                  // addcdiv_(exp_avg, denom, -step_size)
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n4, AdamScalars s, const unsigned* poison = nullptr,
                                              unsigned long long* applied = nullptr, unsigned long long step = 0)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (poison && *poison) return;   // a cross-queue gate timed out: the gradients may be incomplete, keep the parameters
    if (applied && i == 0) *applied = step;   // "optimizer step number `step` was applied": the host rolls its counter back to it after a time-out
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float pe = pp[j], me = mm[j], ve = vv[j];
        adam_element(pe, gg[j], me, ve, s);
        pp[j] = pe; mm[j] = me; vv[j] = ve;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
}

// AdamW{amsgrad: true} (opt.rs:20-27, 45-53 -> libtorch AdamW::step with amsgrad): max_exp_avg_sq = max(max_exp_avg_sq, exp_avg_sq) and
// the denominator uses the running maximum; everything else as adam_element.
__device__ __forceinline__ void adam_element_amsgrad(float& p, float g, float& m, float& v, float& vmax, const AdamScalars& s)
{
#pragma clang fp contract(off)
    p = p * s.wd_mul;
    const float mg = g * s.omb1;
    m = m * s.b1 + mg;
    const float gg = s.omb2 * g * g;
    v = v * s.b2 + gg;
    vmax = fmaxf(vmax, v);                        // torch::max_out(max_exp_avg_sq, exp_avg_sq, max_exp_avg_sq)
    const float denom = __fsqrt_rn(vmax) / s.sqrt_bc2 + s.eps;
    const float upd = s.neg_step * m / denom;
    p = p + upd;
}

__global__ __launch_bounds__(256) void k_adam_amsgrad(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                      float* __restrict__ vmax, size_t n4, AdamScalars s, const unsigned* poison = nullptr,
                                                      unsigned long long* applied = nullptr, unsigned long long step = 0)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (poison && *poison) return;
    if (applied && i == 0) *applied = step;
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], xx = reinterpret_cast<f32x4*>(vmax)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float pe = pp[j], me = mm[j], ve = vv[j], xe = xx[j];
        adam_element_amsgrad(pe, gg[j], me, ve, xe, s);
        pp[j] = pe; mm[j] = me; vv[j] = ve; xx[j] = xe;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(vmax)[i] = xx;
}

// util.rs:31-45 track: dest = tau * src + (1 - tau) * dest, every operation rounded on its own (see adam_element)
__device__ __forceinline__ float track_element(float src, float dst, float tau, float omt)
{
#pragma clang fp contract(off)
    const float x = tau * src;
    const float y = omt * dst;
    return x + y;
}

__global__ __launch_bounds__(256) void k_track(float* __restrict__ dst, const float* __restrict__ src, size_t n4, float tau,
                                               float omt, const unsigned* poison = nullptr)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (poison && *poison) return;   // a cross-queue gate timed out: the update this soft update belongs to was skipped, and so is it
    f32x4 d = reinterpret_cast<f32x4*>(dst)[i], s = reinterpret_cast<const f32x4*>(src)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = track_element(s[j], d[j], tau, omt);
    reinterpret_cast<f32x4*>(dst)[i] = d;
}

__global__ __launch_bounds__(256) void k_scale(float* __restrict__ p, size_t n4, float s)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 d = reinterpret_cast<f32x4*>(p)[i];
    d *= s;
    reinterpret_cast<f32x4*>(p)[i] = d;
}


// Adam scalars on the host in double, entering the f32 element kernel as f32 (libtorch Adam::step)
inline AdamScalars adam_scalars_for(bool adamw, double lr, double beta1, double beta2, double eps, double wd, uint64_t step)
{
    // opt.rs:35: tch nn::Adam::default() -> beta1 .9, beta2 .999, wd 0, eps 1e-8; AdamW: opt.rs:38-55
    const double b1 = adamw ? beta1 : 0.9, b2 = adamw ? beta2 : 0.999, e = adamw ? eps : 1e-8, w = adamw ? wd : 0.0;
    const double bc1 = 1.0 - std::pow(b1, (double)step), bc2 = 1.0 - std::pow(b2, (double)step);
    AdamScalars s;
    s.b1 = (float)b1; s.omb1 = (float)(1.0 - b1); s.b2 = (float)b2; s.omb2 = (float)(1.0 - b2);
    s.sqrt_bc2 = (float)std::sqrt(bc2); s.eps = (float)e; s.neg_step = (float)(-(lr / bc1));
    s.wd_mul = (float)(1.0 - lr * w);
    return s;
}

inline int32_t launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n_floats, const AdamScalars& s)
{
    const size_t n4 = n_floats / 4;
    BDR_HIP(step_launch(st, true, k_adam, dim3((unsigned)((n4 + 255) / 256)), dim3(256), p, g, m, v, n4, s, (const unsigned*)nullptr, (unsigned long long*)nullptr, 0ull));
    return BDR_OK;
}
inline int32_t launch_adam_amsgrad(hipStream_t st, float* p, const float* g, float* m, float* v, float* vmax, size_t n_floats, const AdamScalars& s,
                                   const unsigned* poison = nullptr, unsigned long long* applied = nullptr, unsigned long long step = 0)
{
    const size_t n4 = n_floats / 4;
    BDR_HIP(step_launch(st, true, k_adam_amsgrad, dim3((unsigned)((n4 + 255) / 256)), dim3(256), p, g, m, v, vmax, n4, s, poison, applied, step));
    return BDR_OK;
}
inline int32_t launch_track(hipStream_t st, float* dst, const float* src, size_t n_floats, double tau)
{
    const size_t n4 = n_floats / 4;
    BDR_HIP(step_launch(st, false, k_track, dim3((unsigned)((n4 + 255) / 256)), dim3(256), dst, src, n4, (float)tau, (float)(1.0 - tau), (const unsigned*)nullptr));
    return BDR_OK;
}
inline int32_t launch_scale(hipStream_t st, float* p, size_t n_floats, float sc)
{
    const size_t n4 = n_floats / 4;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p, n4, sc);
    BDR_HIP(hipGetLastError());
    return BDR_OK;
}
}  // namespace
