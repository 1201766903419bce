// Host-side driver of the opt-step loops: the reference's Trainer is compiled Rust, so its counterpart
// above the C ABI is compiled C++ (the Python mirror in border_amd/trainer.py binds the same rules for tests and bench.py).
//   Trainer::train_step      border-core/src/trainer.rs:197-228   warm-up rule, opt_interval, opt vs opt_with_record, timer
//   Trainer::train           trainer.rs:267-327                   sample_and_push, train_step, cost record, stop at max_opts
//   Trainer::train_offline   trainer.rs:330-384                   warmup_period = 0, opt_interval = 1
//   Sampler::sample_and_push trainer/sampler.rs:99-144            reset on first use, Policy::sample, step_with_reset, push
//   SimpleStepProcessor      generic_replay_buffer/step_proc.rs:62-137   (prev_obs, act, obs, ...) transitions
// Recorder / evaluator sinks are out of scope: an observer callback receives what the reference would
// store.  Pure host code: the agent, buffer and environment are reached through function tables, so the loop itself runs
// (and is tested) without a GPU.
#include <chrono>
#include <cmath>
#include <vector>

#include "agent_base.hpp"
#include "common.hpp"

using namespace bdr;

namespace {
using Clock = std::chrono::steady_clock;

struct State {
    const bdr_trainer_config* c;
    const bdr_trainer_ops* ops;
    uint64_t env_steps = 0, opt_steps = 0, opt_steps_counter = 0, samples_counter = 0, n_records = 0, n_episodes = 0;
    double timer_for_opt_steps = 0, timer_for_samples = 0, total_opt = 0, total_sample = 0;
    float scalars[128];
    int32_t n_scalars = 0;
};

// trainer.rs:197-228
int32_t train_step(State& s, bool* is_opt, bool* with_record)
{
    const bdr_trainer_config& c = *s.c;
    *is_opt = false; *with_record = false; s.n_scalars = 0;
    if (s.env_steps < c.warmup_period) return BDR_OK;
    if (s.env_steps % c.opt_interval != 0) return BDR_OK;
    const auto t0 = Clock::now();
    if (c.record_agent_info_interval != 0 && (s.opt_steps + 1) % c.record_agent_info_interval == 0) {
        BDR_TRY(s.ops->agent_opt_with_record(s.ops->agent, s.ops->buffer, s.scalars, 128, &s.n_scalars));
        *with_record = true;
        s.n_records += 1;
    } else {
        BDR_TRY(s.ops->agent_opt(s.ops->agent, s.ops->buffer));
    }
    s.opt_steps += 1;
    const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
    s.timer_for_opt_steps += dt; s.total_opt += dt;
    s.opt_steps_counter += 1;
    *is_opt = true;
    return BDR_OK;
}

// trainer.rs:164-181: average_time() / reset_counters() at every record_compute_cost_interval-th opt step
void cost_record(State& s, bdr_trainer_observer obs, void* ctx)
{
    const bdr_trainer_config& c = *s.c;
    if (c.record_compute_cost_interval == 0 || s.opt_steps % c.record_compute_cost_interval != 0) return;
    // `timer.as_millis() as f32 / n as f32`: the accumulated time is truncated to whole milliseconds first (trainer.rs:164-174)
    const float avr[2] = {s.opt_steps_counter ? (float)std::floor(1000.0 * s.timer_for_opt_steps) / (float)s.opt_steps_counter : -1.0f,
                          s.samples_counter ? (float)std::floor(1000.0 * s.timer_for_samples) / (float)s.samples_counter : -1.0f};
    if (obs) obs(ctx, s.env_steps, s.opt_steps, BDR_TRAINER_EVENT_COST, avr, 2);
    s.timer_for_opt_steps = 0; s.opt_steps_counter = 0; s.timer_for_samples = 0; s.samples_counter = 0;
}

void fill_stats(const State& s, bdr_trainer_stats* out)
{
    if (!out) return;
    out->env_steps = s.env_steps; out->opt_steps = s.opt_steps; out->n_records = s.n_records; out->n_episodes = s.n_episodes;
    out->opt_seconds = s.total_opt; out->sample_seconds = s.total_sample;
}

int32_t check(const bdr_trainer_config* c, const bdr_trainer_ops* ops)
{
    BDR_REQUIRE(c && ops, "null argument");
    BDR_REQUIRE(c->max_opts >= 1, "max_opts must be >= 1");
    BDR_REQUIRE(c->opt_interval >= 1, "opt_interval must be >= 1");
    BDR_REQUIRE(ops->agent_set_train && ops->agent_opt && ops->agent_opt_with_record, "agent function table is incomplete");
    return BDR_OK;
}

// defaults: the library's own handles
int32_t d_set_train(void* a, int32_t on) { return bdr_agent_set_train((bdr_agent*)a, on); }
// Policy::sample of the handle's kind: i64 actions for DQN / IQN, f32 action rows for SAC (sac/base.rs:215-225)
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
// the same two for device-resident observations (bdr_env_vtable::obs_on_device)
int32_t d_sample_dev(void* a, uint64_t n, const void* obs_dev, uint64_t stride, void* act)
{
    bdr_agent* ag = (bdr_agent*)a;
    if (ag && !strcmp(ag->kind(), "sac")) return bdr_sac_sample_device(ag, n, obs_dev, stride, (float*)act);
    return bdr_agent_sample_device(ag, n, obs_dev, stride, (int64_t*)act, nullptr);
}
int32_t d_push_dev(void* b, uint64_t n, const void* obs_dev, uint64_t os, const void* act, const void* next_dev, uint64_t ns, const float* rew,
                   const int8_t* term, const int8_t* trunc)
{
    return bdr_replay_push_device((bdr_replay*)b, n, obs_dev, os, act, next_dev, ns, rew, term, trunc);
}
}  // namespace

namespace bdr {
// the GPU of the library's own ring behind a function table (bdr_trainer_ops_default); false: the caller's own buffer object
bool default_buffer_device(const bdr_trainer_ops* t, int* device)
{
    if (!t || !t->buffer || t->buffer_push_device != d_push_dev) return false;
    *device = ((const bdr_replay*)t->buffer)->device;
    return true;
}
}  // namespace bdr

extern "C" {

void bdr_trainer_config_default(bdr_trainer_config* c)   // trainer/config.rs:68-87 (intervals the loops use; 0 = never)
{
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->max_opts = 0; c->opt_interval = 1; c->warmup_period = 0;
    c->record_agent_info_interval = 0; c->record_compute_cost_interval = 0;
}

void bdr_trainer_ops_default(bdr_trainer_ops* ops, bdr_agent* agent, bdr_replay* buffer)
{
    if (!ops) return;
    ops->agent = agent; ops->buffer = buffer;
    ops->agent_set_train = d_set_train; ops->agent_sample = d_sample; ops->agent_opt = d_opt;
    ops->agent_opt_with_record = d_opt_rec; ops->buffer_push = d_push;
    ops->agent_sample_device = d_sample_dev; ops->buffer_push_device = d_push_dev;
}

int32_t bdr_trainer_train(const bdr_trainer_config* c, const bdr_trainer_ops* ops, const bdr_env_vtable* env,
                          bdr_trainer_observer obs, void* obs_ctx, bdr_trainer_stats* out)
{
    BDR_TRY(check(c, ops));
    BDR_REQUIRE(env && env->reset && env->step_with_reset, "environment function table is incomplete");
    BDR_REQUIRE(ops->agent_sample && ops->buffer_push, "the online loop needs agent_sample and buffer_push");
    BDR_REQUIRE(c->obs_row_bytes > 0 && c->act_row_bytes > 0, "obs_row_bytes / act_row_bytes must be set");
    const bool dev = env->obs_on_device != 0;   // the environment's observations live in HBM: act and push through the *_device entries
    BDR_REQUIRE(!dev || (ops->agent_sample_device && ops->buffer_push_device), "a device-resident environment needs agent_sample_device and buffer_push_device");
    State s; s.c = c; s.ops = ops;
    // Sampler state (sampler.rs:61-75) + SimpleStepProcessor::prev_obs (step_proc.rs:86-101).  The sampler's prev_obs and the step
    // processor's are the same observation at the top of every iteration (both become init_obs after a terminal step, obs otherwise):
    // one buffer, host or device
    ObsRow prev, obs_new, init_obs;
    BDR_TRY(prev.init(dev, env->device, c->obs_row_bytes)); BDR_TRY(obs_new.init(dev, env->device, c->obs_row_bytes)); BDR_TRY(init_obs.init(dev, env->device, c->obs_row_bytes));
    std::vector<uint8_t> act(c->act_row_bytes);
    bool have_prev = false;
    BDR_TRY(ops->agent_set_train(ops->agent, 1));   // trainer.rs:283
    for (;;) {
        const auto t0 = Clock::now();
        // ---- Sampler::sample_and_push (sampler.rs:99-144)
        if (!have_prev) {
            BDR_TRY(env->reset(env->ctx, prev.p()));    // (+ step_processor.reset(prev_obs.clone()))
            have_prev = true;
        }
        if (dev) BDR_TRY(ops->agent_sample_device(ops->agent, 1, prev.p(), c->obs_row_bytes, act.data()));
        else BDR_TRY(ops->agent_sample(ops->agent, 1, prev.p(), act.data()));
        float reward = 0; int8_t term = 0, trunc = 0;
        BDR_TRY(env->step_with_reset(env->ctx, act.data(), obs_new.p(), &reward, &term, &trunc, init_obs.p()));
        const bool is_done = term == 1 || trunc == 1;   // step.rs:136-138
        // SimpleStepProcessor::process (step_proc.rs:103-137): (prev_obs, act, obs, reward, flags); next transition starts
        // from obs, or from init_obs after a terminal step
        if (dev) BDR_TRY(ops->buffer_push_device(ops->buffer, 1, prev.p(), c->obs_row_bytes, act.data(), obs_new.p(), c->obs_row_bytes, &reward, &term, &trunc));
        else BDR_TRY(ops->buffer_push(ops->buffer, 1, prev.p(), act.data(), obs_new.p(), &reward, &term, &trunc));
        prev.swap(is_done ? init_obs : obs_new);        // prev_obs = init_obs / obs (sampler.rs:137-141, step_proc.rs:131-135)
        if (is_done) s.n_episodes += 1;
        const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
        s.timer_for_samples += dt; s.total_sample += dt;
        s.samples_counter += 1;
        s.env_steps += 1;
        // ---- train_step + records
        bool is_opt = false, with_record = false;
        BDR_TRY(train_step(s, &is_opt, &with_record));
        if (obs) obs(obs_ctx, s.env_steps, s.opt_steps, is_opt ? (with_record ? BDR_TRAINER_EVENT_OPT_RECORD : BDR_TRAINER_EVENT_OPT) : BDR_TRAINER_EVENT_SKIP,
                     s.scalars, s.n_scalars);
        cost_record(s, obs, obs_ctx);
        if (s.opt_steps == c->max_opts) break;           // trainer.rs:323-325
    }
    fill_stats(s, out);
    return BDR_OK;
}

int32_t bdr_trainer_train_offline(const bdr_trainer_config* c_in, const bdr_trainer_ops* ops, bdr_trainer_observer obs, void* obs_ctx,
                                  bdr_trainer_stats* out)
{
    BDR_TRY(check(c_in, ops));
    bdr_trainer_config c = *c_in;
    c.warmup_period = 0; c.opt_interval = 1;            // trainer.rs:345-346
    State s; s.c = &c; s.ops = ops;
    BDR_TRY(ops->agent_set_train(ops->agent, 1));
    for (;;) {
        s.env_steps += 1;
        bool is_opt = false, with_record = false;
        BDR_TRY(train_step(s, &is_opt, &with_record));
        if (obs) obs(obs_ctx, s.env_steps, s.opt_steps, is_opt ? (with_record ? BDR_TRAINER_EVENT_OPT_RECORD : BDR_TRAINER_EVENT_OPT) : BDR_TRAINER_EVENT_SKIP,
                     s.scalars, s.n_scalars);
        cost_record(s, obs, obs_ctx);
        if (s.opt_steps == c.max_opts) break;
    }
    fill_stats(s, out);
    return BDR_OK;
}

}  // extern "C"
