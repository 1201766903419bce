// border-async-trainer on MI355X: the compiled counterpart of
//   AsyncTrainer::{train, train_step, sync, update_replay_buffer}   border-async-trainer/src/async_trainer/base.rs:204-222,268-284,299-388
//   Actor::run                                                       actor/base.rs:120-178
//   ActorManager::run (actor threads, the shared model info)         actor_manager/base.rs:112-186
//   ReplayBufferProxy::push (n_buffer batching, try_send)            replay_buffer_proxy.rs:52-72
//   SyncModel                                                        sync_model.rs:2-13
//   AsyncTrainStat / ActorStat                                       async_trainer/stat.rs, actor/stat.rs
//
// Mapping (north_star): one rank per GPU runs ONE learner (its agent + its local replay shard in HBM) and its own actors on
// that GPU; the actors' transitions never leave the rank.  The learner -> actors model channel of the reference (a
// NamedTensors message over a crossbeam channel, one host copy per actor) is a device-resident mailbox here
// (bdr_model_mailbox): the learner's publish is one device-to-device copy enqueued on its own stream, an actor's sync_model is
// one copy enqueued on the actor's stream, ordered by two events - neither side synchronises with the host.  Across GPUs the
// `exchange` hook runs at every sync point before the local publish (bdr_agent_allreduce_params: RCCL over xGMI).
//
// The loop itself is host code behind function tables (like csrc/trainer.hip), so its rules are testable without a GPU:
// agents / buffers / environments / the mailbox may be the library's handles (bdr_async_ops_default) or the caller's objects.
// The two forwarding threads of ActorManager (handle_message: channel -> channel, run_model_info_loop: channel -> shared slot)
// are plumbing between the reference's two structs and are folded away: actors send into the trainer's bounded queue
// (bounded(1000), actor_manager/base.rs:140) and read the shared slot the trainer writes.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "agent_base.hpp"
#include "common.hpp"

using namespace bdr;

namespace bdr {
float* agent_arena(bdr_agent* a, int which, size_t* n_floats, hipStream_t* stream, int* device);   // agent_api.hip
bool default_buffer_device(const bdr_trainer_ops* t, int* device);                                  // trainer.hip
}

// ---- device-resident SyncModel::ModelInfo mailbox --------------------------------------------------------------------------
struct bdr_model_mailbox {
    int32_t device = 0;
    uint64_t n_floats = 0;
    float* snap = nullptr;            // the published parameter arena (kernel layout; identical on both sides)
    hipEvent_t written = nullptr;     // recorded on the publisher's stream after the copy into `snap`
    std::vector<hipEvent_t> read;     // per reader: recorded on the reader's stream after its copy out of `snap`
    std::vector<char> read_pending;
    uint64_t n_opts = 0;              // model_info().0
    bool valid = false;
    std::mutex mu;                    // serialises ENQUEUEING (not execution): event record / wait order == lock order
};

extern "C" {

int32_t bdr_model_mailbox_create(int32_t device, uint64_t n_floats, uint32_t n_readers, bdr_model_mailbox** out)
{
    BDR_REQUIRE(out && n_floats > 0 && n_readers >= 1 && n_readers <= 1024, "bad argument");
    BDR_TRY(ensure_device(device));
    bdr_model_mailbox* m = new bdr_model_mailbox();
    m->device = device; m->n_floats = n_floats;
    hipError_t e = hipMalloc((void**)&m->snap, n_floats * 4);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->written, hipEventDisableTiming);
    m->read.assign(n_readers, nullptr); m->read_pending.assign(n_readers, 0);
    for (uint32_t i = 0; i < n_readers && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&m->read[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        int32_t st = fail(BDR_ERR_HIP, "model mailbox allocation failed: %s", hipGetErrorString(e));
        (void)hipFree(m->snap);
        delete m;
        return st;
    }
    *out = m;
    return BDR_OK;
}

int32_t bdr_model_mailbox_destroy(bdr_model_mailbox* m)
{
    if (!m) return BDR_OK;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(m->snap);
    if (m->written) (void)hipEventDestroy(m->written);
    for (auto e : m->read) if (e) (void)hipEventDestroy(e);
    delete m;
    return BDR_OK;
}

// AsyncTrainer::sync (async_trainer/base.rs:268-272): model_info() -> send.  Asynchronous on the learner's stream.
int32_t bdr_agent_publish_model(bdr_agent* a, int32_t which, bdr_model_mailbox* m, uint64_t n_opts)
{
    BDR_REQUIRE(a && m, "null argument");
    size_t n = 0; hipStream_t st = nullptr; int dev = 0;
    float* p = agent_arena(a, which, &n, &st, &dev);
    BDR_REQUIRE(p && n == m->n_floats && dev == m->device, "mailbox does not match the agent's parameter arena");
    BDR_HIP(hipSetDevice(dev));
    std::lock_guard<std::mutex> l(m->mu);
    for (size_t i = 0; i < m->read.size(); ++i)      // WAR: readers still copying the previous snapshot
        if (m->read_pending[i]) { BDR_HIP(hipStreamWaitEvent(st, m->read[i], 0)); m->read_pending[i] = 0; }
    BDR_HIP(hipMemcpyAsync(m->snap, p, n * 4, hipMemcpyDeviceToDevice, st));
    BDR_HIP(hipEventRecord(m->written, st));
    m->n_opts = n_opts; m->valid = true;
    return BDR_OK;
}

// Actor::sync_model (actor/base.rs:104-118): `if model_info.0 > *n_opt_steps { agent.sync_model(..) }`; `first` = sync_model_first
// (:98-102, unconditional).  Asynchronous on the actor's stream.
int32_t bdr_agent_sync_model_from(bdr_agent* a, int32_t which, bdr_model_mailbox* m, uint32_t reader, int32_t first,
                                  uint64_t* n_opts_inout, int32_t* updated)
{
    BDR_REQUIRE(a && m && n_opts_inout, "null argument");
    BDR_REQUIRE(reader < m->read.size(), "reader index out of range");
    if (updated) *updated = 0;
    size_t n = 0; hipStream_t st = nullptr; int dev = 0;
    float* p = agent_arena(a, which, &n, &st, &dev);
    BDR_REQUIRE(p && n == m->n_floats && dev == m->device, "mailbox does not match the agent's parameter arena");
    std::lock_guard<std::mutex> l(m->mu);
    if (!m->valid) return first ? fail(BDR_ERR_INVALID, "no model has been published yet") : BDR_OK;
    if (!first && m->n_opts <= *n_opts_inout) return BDR_OK;
    BDR_HIP(hipSetDevice(dev));
    BDR_HIP(hipStreamWaitEvent(st, m->written, 0));
    BDR_HIP(hipMemcpyAsync(p, m->snap, n * 4, hipMemcpyDeviceToDevice, st));
    BDR_HIP(hipEventRecord(m->read[reader], st));
    m->read_pending[reader] = 1;
    *n_opts_inout = m->n_opts;
    if (updated) *updated = 1;
    return BDR_OK;
}

}  // extern "C"

// ---- the loops -----------------------------------------------------------------------------------------------------------
namespace {
using Clock = std::chrono::steady_clock;

// Device rows of the messages of device-resident actors, recycled: an actor takes a pair of [n_buffer][obs_row_bytes] buffers when it
// starts a message, the learner hands them back when the message has been pushed (bdr_replay_push_device has synchronised by then).
// hipMalloc / hipFree per message would stall the learner's device (hipFree synchronises the whole device).
struct DevRowPool {
    struct Pair { int device; size_t bytes; uint8_t* a; uint8_t* b; };
    std::mutex mu;
    std::vector<Pair> free_;
    bool take(int device, size_t bytes, uint8_t** a, uint8_t** b)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].device == device && free_[i].bytes == bytes) { *a = free_[i].a; *b = free_[i].b; free_.erase(free_.begin() + (long)i); return true; }
        }
        *a = *b = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)a, bytes) != hipSuccess) return false;
        if (hipMalloc((void**)b, bytes) != hipSuccess) { (void)hipFree(*a); *a = nullptr; return false; }
        return true;
    }
    void give(int device, size_t bytes, uint8_t* a, uint8_t* b)
    {
        std::lock_guard<std::mutex> l(mu);
        free_.push_back(Pair{device, bytes, a, b});
    }
    ~DevRowPool() { for (auto& p : free_) { (void)hipSetDevice(p.device); (void)hipFree(p.a); (void)hipFree(p.b); } }
};

// PushedItemMessage { id, pushed_items } (messages.rs): n transitions packed field by field
struct Message {
    uint32_t id = 0;
    uint64_t n = 0;
    std::vector<uint8_t> obs, act, next_obs;
    std::vector<float> reward;
    std::vector<int8_t> term, trunc;
    // device-resident environments (bdr_env_vtable::obs_on_device): the observation rows of the message stay in HBM -
    // [n_buffer][obs_row_bytes] each, taken from the run's pool by the actor, pushed with buffer_push_device and handed back by the learner
    uint8_t* d_obs = nullptr; uint8_t* d_next = nullptr; int device = 0; size_t d_bytes = 0; DevRowPool* pool = nullptr;
    Message() = default;
    Message(const Message&) = delete; Message& operator=(const Message&) = delete;
    Message(Message&& o) noexcept { *this = std::move(o); }
    Message& operator=(Message&& o) noexcept
    {
        if (this != &o) {
            release();
            id = o.id; n = o.n; obs = std::move(o.obs); act = std::move(o.act); next_obs = std::move(o.next_obs);
            reward = std::move(o.reward); term = std::move(o.term); trunc = std::move(o.trunc);
            d_obs = o.d_obs; d_next = o.d_next; device = o.device; d_bytes = o.d_bytes; pool = o.pool; o.d_obs = o.d_next = nullptr; o.n = 0;
        }
        return *this;
    }
    ~Message() { release(); }
    void release()
    {
        if (!d_obs && !d_next) return;
        if (pool) pool->give(device, d_bytes, d_obs, d_next);
        else { (void)hipSetDevice(device); (void)hipFree(d_obs); (void)hipFree(d_next); }
        d_obs = d_next = nullptr;
    }
};

// crossbeam bounded(cap): try_send fails when full (replay_buffer_proxy.rs:63-68), try_iter drains what is there
struct Channel {
    std::mutex mu;
    std::deque<Message> q;
    size_t cap = 1000;
    bool try_send(Message&& m)
    {
        std::lock_guard<std::mutex> l(mu);
        if (q.size() >= cap) return false;
        q.push_back(std::move(m));
        return true;
    }
    void drain(std::vector<Message>& out)
    {
        std::lock_guard<std::mutex> l(mu);
        while (!q.empty()) { out.push_back(std::move(q.front())); q.pop_front(); }
    }
};

struct Shared {
    const bdr_async_trainer_config* c;
    DevRowPool rows;                   // (declared before the channel: messages still queued at the end hand their rows back first)
    Channel ch;
    std::atomic<bool> stop{false};
    std::atomic<bool> model_ready{false};
    std::mutex obs_mu;                 // the observer is never called concurrently
    bdr_async_observer observer = nullptr;
    void* observer_ctx = nullptr;
    std::mutex err_mu;
    int32_t err = BDR_OK;
    char err_msg[512] = "";
    void fail_from(const char* who, uint32_t id)
    {
        std::lock_guard<std::mutex> l(err_mu);
        if (err == BDR_OK) { err = BDR_ERR_INVALID; snprintf(err_msg, sizeof err_msg, "%s %u: %s", who, id, bdr_last_error()); }
        stop.store(true);
    }
    void notify(uint32_t actor, uint64_t a, uint64_t b, int32_t ev, const float* s, int32_t n)
    {
        if (!observer) return;
        std::lock_guard<std::mutex> l(obs_mu);
        observer(observer_ctx, actor, a, b, ev, s, n);
    }
};

// Actor::run (actor/base.rs:120-178) with Sampler::sample_and_push (border-core/src/trainer/sampler.rs:99-144),
// SimpleStepProcessor (generic_replay_buffer/step_proc.rs:62-137) and ReplayBufferProxy::push (replay_buffer_proxy.rs:52-72)
void actor_run(Shared* sh, const bdr_actor_ops* ops, uint32_t id, bdr_actor_stat* stat)
{
    const bdr_async_trainer_config& c = *sh->c;
    const auto t_start = Clock::now();
    uint64_t env_steps = 0, n_opt_steps = 0;
    const bool dev = ops->env.obs_on_device != 0;   // observations stay in HBM from the environment to the learner's ring
    // (the sampler's prev_obs and the step processor's are the same observation at the top of every iteration: one buffer)
    ObsRow prev, obs_new, init_obs;
    std::vector<uint8_t> act(c.act_row_bytes);
    Message buf;
    auto reset_buf = [&]() -> bool {
        buf = Message();
        buf.id = id;
        buf.act.reserve(c.n_buffer * c.act_row_bytes);
        if (!dev) { buf.obs.reserve(c.n_buffer * c.obs_row_bytes); buf.next_obs.reserve(c.n_buffer * c.obs_row_bytes); return true; }
        buf.device = ops->env.device; buf.d_bytes = c.n_buffer * c.obs_row_bytes; buf.pool = &sh->rows;
        if (!sh->rows.take(buf.device, buf.d_bytes, &buf.d_obs, &buf.d_next)) {
            (void)fail(BDR_ERR_HIP, "actor %u: device rows of a message (2 x %llu bytes) could not be allocated", id, (unsigned long long)(c.n_buffer * c.obs_row_bytes));
            return false;
        }
        return true;
    };
    // The rows of a transition are copied on the actor's own stream and the copy is COMPLETE before the loop goes on: a same-device
    // hipMemcpy does not wait on the host, and the null stream orders nothing against the non-blocking streams of the environment
    // (bdr_atari_prep), the agent and the learner's ring - the next environment step overwrites the source, the learner reads the copy.
    hipStream_t copy_st = nullptr;
    auto finish = [&]() {
        if (copy_st) { (void)hipSetDevice(ops->env.device); (void)hipStreamSynchronize(copy_st); (void)hipStreamDestroy(copy_st); copy_st = nullptr; }
        if (stat) { stat->env_steps = env_steps; stat->duration_s = std::chrono::duration<double>(Clock::now() - t_start).count(); }
    };
    if (prev.init(dev, ops->env.device, c.obs_row_bytes) != BDR_OK || obs_new.init(dev, ops->env.device, c.obs_row_bytes) != BDR_OK ||
        init_obs.init(dev, ops->env.device, c.obs_row_bytes) != BDR_OK || !reset_buf()) { sh->fail_from("actor", id); finish(); return; }
    if (dev && (hipSetDevice(ops->env.device) != hipSuccess || hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking) != hipSuccess)) {
        (void)fail(BDR_ERR_HIP, "actor %u: the copy stream could not be created", id);
        sh->fail_from("actor", id); finish(); return;
    }
    // "Waits and syncs the initial model" (:148-153)
    while (!sh->model_ready.load() && !sh->stop.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (sh->stop.load()) { finish(); return; }
    if (ops->sync_model(ops->agent, ops->mailbox, id, 1, &n_opt_steps, nullptr) != BDR_OK) { sh->fail_from("actor", id); finish(); return; }
    if (stat) stat->n_syncs += 1;
    sh->notify(id, env_steps, n_opt_steps, BDR_ASYNC_EVENT_ACTOR_SYNC, nullptr, 0);
    if (ops->agent_set_train(ops->agent, 1) != BDR_OK) { sh->fail_from("actor", id); finish(); return; }   // :156
    bool have_prev = false;
    for (;;) {
        // "Check model update and synchronize" (:161-167)
        int32_t updated = 0;
        if (ops->sync_model(ops->agent, ops->mailbox, id, 0, &n_opt_steps, &updated) != BDR_OK) { sh->fail_from("actor", id); break; }
        if (updated) { if (stat) stat->n_syncs += 1; sh->notify(id, env_steps, n_opt_steps, BDR_ASYNC_EVENT_ACTOR_SYNC, nullptr, 0); }
        // ---- sampler.sample_and_push(&mut agent, &mut buffer) (:170)
        if (!have_prev) {
            if (ops->env.reset(ops->env.ctx, prev.p()) != BDR_OK) { sh->fail_from("actor env", id); break; }
            have_prev = true;
        }
        if ((dev ? ops->agent_sample_device(ops->agent, 1, prev.p(), c.obs_row_bytes, act.data()) : ops->agent_sample(ops->agent, 1, prev.p(), act.data())) != BDR_OK) {
            sh->fail_from("actor", id); break;
        }
        float reward = 0; int8_t term = 0, trunc = 0;
        if (ops->env.step_with_reset(ops->env.ctx, act.data(), obs_new.p(), &reward, &term, &trunc, init_obs.p()) != BDR_OK) { sh->fail_from("actor env", id); break; }
        const bool is_done = term == 1 || trunc == 1;
        // ReplayBufferProxy::push: buffer the item; at n_buffer items swap the Vec out and try_send it
        if (dev) {
            if (hipMemcpyAsync(buf.d_obs + buf.n * c.obs_row_bytes, prev.p(), c.obs_row_bytes, hipMemcpyDeviceToDevice, copy_st) != hipSuccess ||
                hipMemcpyAsync(buf.d_next + buf.n * c.obs_row_bytes, obs_new.p(), c.obs_row_bytes, hipMemcpyDeviceToDevice, copy_st) != hipSuccess ||
                hipStreamSynchronize(copy_st) != hipSuccess) {
                (void)fail(BDR_ERR_HIP, "actor %u: device copy of a transition's rows failed", id);
                sh->fail_from("actor", id); break;
            }
        } else {
            const uint8_t* po = static_cast<const uint8_t*>(prev.p()); const uint8_t* pn = static_cast<const uint8_t*>(obs_new.p());
            buf.obs.insert(buf.obs.end(), po, po + c.obs_row_bytes);
            buf.next_obs.insert(buf.next_obs.end(), pn, pn + c.obs_row_bytes);
        }
        buf.act.insert(buf.act.end(), act.begin(), act.end());
        buf.reward.push_back(reward); buf.term.push_back(term); buf.trunc.push_back(trunc);
        buf.n += 1;
        prev.swap(is_done ? init_obs : obs_new);   // prev_obs = init_obs / obs for the sampler and the step processor alike
        if (buf.n == c.n_buffer) {
            Message m = std::move(buf);
            if (!reset_buf()) { sh->fail_from("actor", id); break; }
            if (!sh->ch.try_send(std::move(m))) {   // Err(SendMsgForPush): the reference's actor thread panics on the unwrap (:170)
                (void)fail(BDR_ERR_INVALID, "ReplayBufferProxy::push: the bounded channel (%llu messages) is full (SendMsgForPush)",
                           (unsigned long long)c.channel_capacity);
                sh->fail_from("actor", id);
                break;
            }
        }
        env_steps += 1;
        if (sh->stop.load()) break;   // "Stop sampling loop" (:174-180)
    }
    finish();
}

}  // namespace

extern "C" {

void bdr_async_trainer_config_default(bdr_async_trainer_config* c)   // async_trainer/config.rs:101-112, actor_manager/config.rs:12-16
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->max_opts = 10; c->record_compute_cost_interval = 5000; c->record_agent_info_interval = 5000;
    c->sync_interval = 100; c->warmup_period = 10000; c->n_buffer = 100; c->channel_capacity = 1000; c->warmup_sleep_ms = 100;
}

// AsyncTrainer::train (async_trainer/base.rs:299-388) + ActorManager::run / stop_and_join (actor_manager/base.rs:112-215)
int32_t bdr_async_train(const bdr_async_trainer_config* c, const bdr_learner_ops* L, const bdr_actor_ops* actors, uint32_t n_actors,
                        bdr_async_observer observer, void* observer_ctx, bdr_async_stats* out, bdr_actor_stat* actor_stats)
{
    BDR_REQUIRE(c && L && actors && n_actors >= 1, "null argument");
    BDR_REQUIRE(c->max_opts >= 1 && c->sync_interval >= 1 && c->n_buffer >= 1 && c->channel_capacity >= 1, "max_opts, sync_interval, n_buffer, channel_capacity must be >= 1");
    BDR_REQUIRE(c->obs_row_bytes > 0 && c->act_row_bytes > 0, "obs_row_bytes / act_row_bytes must be set");
    BDR_REQUIRE(L->t.agent_set_train && L->t.agent_opt && L->t.agent_opt_with_record && L->t.buffer_push && L->buffer_len && L->publish_model,
                "learner function table is incomplete");
    for (uint32_t i = 0; i < n_actors; ++i)
        BDR_REQUIRE(actors[i].agent_set_train && actors[i].agent_sample && actors[i].sync_model && actors[i].env.reset && actors[i].env.step_with_reset &&
                    (!actors[i].env.obs_on_device || actors[i].agent_sample_device), "actor %u: function table is incomplete", i);
    // device-resident observations are pushed without leaving HBM (bdr_replay_push_device): the rows must live on the learner ring's GPU.
    // Checked here, not at the first message of a run that has already started its actors.
    int ring_dev = -1;
    if (default_buffer_device(&L->t, &ring_dev))
        for (uint32_t i = 0; i < n_actors; ++i)
            BDR_REQUIRE(!actors[i].env.obs_on_device || actors[i].env.device == ring_dev,
                        "actor %u keeps its observations on GPU %d, the learner's replay buffer lives on GPU %d: device-resident rows are pushed without a "
                        "host copy and must be on the buffer's GPU (run this actor's environment on GPU %d, or with obs_on_device = 0)", i,
                        (int)actors[i].env.device, ring_dev, ring_dev);
    Shared sh;
    sh.c = c; sh.ch.cap = c->channel_capacity; sh.observer = observer; sh.observer_ctx = observer_ctx;
    if (actor_stats) memset(actor_stats, 0, sizeof(bdr_actor_stat) * n_actors);
    std::vector<std::thread> threads;
    for (uint32_t i = 0; i < n_actors; ++i) threads.emplace_back(actor_run, &sh, &actors[i], i, actor_stats ? &actor_stats[i] : nullptr);
    auto stop_and_join = [&]() { sh.stop.store(true); for (auto& t : threads) if (t.joinable()) t.join(); };
    // Multi-rank runs: a rank that fails between two sync points tells its peers so at what would have been its next sync point
    // (L->agree, one MIN all-reduce of an ok flag per sync point), unless the failure is the communication itself.
    bool comm_failed = false, told_peers = false;
    auto tell_peers = [&]() {
        if (!L->agree || comm_failed || told_peers) return;
        told_peers = true;
        int32_t all = 0;
        (void)L->agree(L->exchange_ctx, 0, &all);
    };
#define LEARNER_TRY(expr)                                                                  \
    do { int32_t s__ = (expr); if (s__ != BDR_OK) { std::string m__ = bdr_last_error(); tell_peers(); stop_and_join(); return fail(s__, "%s", m__.c_str()); } } while (0)

    uint64_t opt_steps = 0, samples_total = 0, samples_counter = 0, opt_steps_counter = 0, n_records = 0, n_syncs = 0, n_messages = 0;
    double timer_for_samples = 0, timer_for_opt_steps = 0;
    float scalars[128]; int32_t n_scalars = 0;
    LEARNER_TRY(L->t.agent_set_train(L->t.agent, 1));                      // agent.train() (:317)
    const auto time_total = Clock::now();
    auto sync = [&]() -> int32_t {                                         // AsyncTrainer::sync (:268-272)
        if (L->agree) {                                                     // every rank enters the collective, or none does
            int32_t all = 1;
            const int32_t s = L->agree(L->exchange_ctx, 1, &all);
            if (s != BDR_OK) { comm_failed = true; return s; }
            if (!all) { told_peers = true; return fail(BDR_ERR_COMM, "another rank's learner failed: stopping at the sync point of opt step %llu", (unsigned long long)opt_steps); }
        }
        if (L->exchange) {                                                  // cross-GPU averaging first (RCCL)
            const int32_t s = L->exchange(L->exchange_ctx, L->t.agent, opt_steps);
            if (s != BDR_OK) { comm_failed = true; return s; }
        }
        BDR_TRY(L->publish_model(L->t.agent, L->mailbox, opt_steps));
        n_syncs += 1;
        sh.model_ready.store(true);
        sh.notify(UINT32_MAX, samples_total, opt_steps, BDR_ASYNC_EVENT_SYNC, nullptr, 0);
        return BDR_OK;
    };
    auto update_replay_buffer = [&]() -> int32_t {                         // :275-284
        std::vector<Message> msgs;
        sh.ch.drain(msgs);
        for (auto& m : msgs) {
            samples_counter += m.n; samples_total += m.n; n_messages += 1;
            if (m.d_obs) {   // a device-resident actor: HBM -> ring inside the device
                BDR_REQUIRE(L->t.buffer_push_device, "a device-resident actor needs the learner's buffer_push_device");
                BDR_TRY(L->t.buffer_push_device(L->t.buffer, m.n, m.d_obs, c->obs_row_bytes, m.act.data(), m.d_next, c->obs_row_bytes, m.reward.data(), m.term.data(), m.trunc.data()));
            } else {
                BDR_TRY(L->t.buffer_push(L->t.buffer, m.n, m.obs.data(), m.act.data(), m.next_obs.data(), m.reward.data(), m.term.data(), m.trunc.data()));
            }
            sh.notify(m.id, samples_total, opt_steps, BDR_ASYNC_EVENT_PUSH, nullptr, (int32_t)m.n);
        }
        return BDR_OK;
    };
    LEARNER_TRY(sync());                                                   // "Send model info first in AsyncTrainer" (:325-326)
    // "Warmup period" (:328-335)
    for (;;) {
        LEARNER_TRY(update_replay_buffer());
        uint64_t len = 0;
        LEARNER_TRY(L->buffer_len(L->t.buffer, &len));
        if (len >= c->warmup_period) { std::this_thread::sleep_for(std::chrono::milliseconds(c->warmup_sleep_ms)); break; }
        if (sh.stop.load()) break;   // an actor failed
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    // "Starts training loop" (:338-376)
    while (!sh.stop.load()) {
        const auto t0 = Clock::now();
        LEARNER_TRY(update_replay_buffer());
        timer_for_samples += std::chrono::duration<double>(Clock::now() - t0).count();
        // train_step (:204-222)
        uint64_t len = 0;
        LEARNER_TRY(L->buffer_len(L->t.buffer, &len));
        int32_t ev = BDR_ASYNC_EVENT_SKIP;
        n_scalars = 0;
        if (len >= c->warmup_period) {
            const auto t1 = Clock::now();
            if (c->record_agent_info_interval != 0 && (opt_steps + 1) % c->record_agent_info_interval == 0) {
                LEARNER_TRY(L->t.agent_opt_with_record(L->t.agent, L->t.buffer, scalars, 128, &n_scalars));
                ev = BDR_ASYNC_EVENT_OPT_RECORD; n_records += 1;
            } else {
                LEARNER_TRY(L->t.agent_opt(L->t.agent, L->t.buffer));
                ev = BDR_ASYNC_EVENT_OPT;
            }
            opt_steps += 1; opt_steps_counter += 1;
            timer_for_opt_steps += std::chrono::duration<double>(Clock::now() - t1).count();
        }
        sh.notify(UINT32_MAX, samples_total, opt_steps, ev, scalars, n_scalars);
        // post_process (:224-266): evaluation / model saving are recorder & evaluator work (out of scope); "Sync the current model"
        if (opt_steps % c->sync_interval == 0) LEARNER_TRY(sync());
        // average_opt_time / average_sample_time (:190-201, 354-359); as_millis() truncates the accumulated time first
        if (c->record_compute_cost_interval != 0 && opt_steps % c->record_compute_cost_interval == 0) {
            const float avr[2] = {opt_steps_counter ? (float)std::floor(1000.0 * timer_for_opt_steps) / (float)opt_steps_counter : -1.0f,
                                  samples_counter ? (float)std::floor(1000.0 * timer_for_samples) / (float)samples_counter : -1.0f};
            sh.notify(UINT32_MAX, samples_total, opt_steps, BDR_ASYNC_EVENT_COST, avr, 2);
            samples_counter = 0; timer_for_samples = 0; opt_steps_counter = 0; timer_for_opt_steps = 0;
        }
        if (opt_steps == c->max_opts) {                                    // "Finish training" (:368-375)
            sh.stop.store(true);
            std::vector<Message> rest;
            sh.ch.drain(rest);                                             // "Flush channels": dropped, like the reference
            LEARNER_TRY(sync());
            break;
        }
    }
    const double duration = std::chrono::duration<double>(Clock::now() - time_total).count();
    stop_and_join();
    if (sh.err != BDR_OK) tell_peers();   // an actor failed (the loop above ended on its stop flag)
#undef LEARNER_TRY
    if (out) {
        out->samples_total = samples_total; out->opt_steps = opt_steps; out->n_records = n_records; out->n_syncs = n_syncs; out->n_messages = n_messages;
        out->duration_s = duration;
        out->samples_per_sec = (float)((double)samples_total / duration);     // :379-387
        out->opt_per_sec = (float)((double)c->max_opts / duration);
    }
    if (sh.err != BDR_OK) return fail(sh.err, "%s", sh.err_msg);
    return BDR_OK;
}

}  // extern "C"

// ---- default function tables: the library's own handles ----------------------------------------------------------------------
namespace {
struct DefaultCtx { int32_t which; };
int32_t d_set_train(void* a, int32_t on) { return bdr_agent_set_train((bdr_agent*)a, on); }
// Policy::sample of the handle's kind: discrete agents return i64 actions (dqn/base.rs:211-242, iqn/base.rs:204-228), SAC f32 rows
// (sac/base.rs:215-225)
int32_t d_sample(void* a, uint64_t n, const void* obs, void* act)
{
    bdr_agent* ag = (bdr_agent*)a;
    if (ag && !strcmp(ag->kind(), "sac")) return bdr_sac_sample(ag, n, (const float*)obs, (float*)act);
    return bdr_agent_sample(ag, n, obs, (int64_t*)act, nullptr);
}
int32_t d_opt(void* a, void* b) { return bdr_agent_opt((bdr_agent*)a, (bdr_replay*)b); }
int32_t d_opt_rec(void* a, void* b, float* out, int32_t cap, int32_t* n) { return bdr_agent_opt_with_scalars((bdr_agent*)a, (bdr_replay*)b, out, cap, n); }
int32_t d_push(void* b, uint64_t n, const void* obs, const void* act, const void* next_obs, const float* rew, const int8_t* term, const int8_t* trunc)
{
    return bdr_replay_push((bdr_replay*)b, n, obs, act, next_obs, rew, term, trunc);
}
int32_t d_len(void* b, uint64_t* len) { return bdr_replay_len((bdr_replay*)b, len); }
int32_t d_publish(void* a, void* mailbox, uint64_t n_opts) { return bdr_agent_publish_model((bdr_agent*)a, 0, (bdr_model_mailbox*)mailbox, n_opts); }
int32_t d_sync(void* a, void* mailbox, uint32_t reader, int32_t first, uint64_t* n_opts, int32_t* updated)
{
    return bdr_agent_sync_model_from((bdr_agent*)a, 0, (bdr_model_mailbox*)mailbox, reader, first, n_opts, updated);
}
}  // namespace

extern "C" {

namespace {
int32_t d_sample_dev(void* a, uint64_t n, const void* obs_dev, uint64_t stride, void* act)
{
    bdr_agent* ag = (bdr_agent*)a;
    if (ag && !strcmp(ag->kind(), "sac")) return bdr_sac_sample_device(ag, n, obs_dev, stride, (float*)act);
    return bdr_agent_sample_device(ag, n, obs_dev, stride, (int64_t*)act, nullptr);
}
}  // namespace

void bdr_learner_ops_default(bdr_learner_ops* ops, bdr_agent* agent, bdr_replay* buffer, bdr_model_mailbox* mailbox)
{
    if (!ops) return;
    memset(ops, 0, sizeof *ops);
    bdr_trainer_ops_default(&ops->t, agent, buffer);   // the Trainer's table (trainer.hip): the same wrappers, and how bdr_async_train knows the ring is the library's own
    ops->buffer_len = d_len; ops->publish_model = d_publish; ops->mailbox = mailbox;
}

void bdr_actor_ops_default(bdr_actor_ops* ops, bdr_agent* agent, bdr_model_mailbox* mailbox, const bdr_env_vtable* env)
{
    if (!ops) return;
    memset(ops, 0, sizeof *ops);
    ops->agent = agent; ops->mailbox = mailbox;
    ops->agent_set_train = d_set_train; ops->agent_sample = d_sample; ops->sync_model = d_sync;
    ops->agent_sample_device = d_sample_dev;
    if (env) ops->env = *env;
}

}  // extern "C"
