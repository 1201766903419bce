/*
 * border_amd.h -- C ABI of the MI355X-native opt-step engine that sits behind border-core's
 * Agent / ReplayBufferBase / ExperienceBufferBase / SyncModel traits.
 *
 * The reference has no FFI of its own on this path (its only native boundary is
 * tch -> torch-sys -> libtorch, which this library replaces), so every entry point below is
 * shaped 1:1 after the Rust trait method it implements; the file:line of that method
 * (relative to the reference tree) is cited next to each declaration.  INTEGRATION.md shows
 * the Rust shim (`impl Agent for AmdDqn`, `impl ReplayBufferBase for AmdReplayBuffer`) a
 * maintainer would add on top of these symbols.
 *
 * Conventions
 *   - plain C types only; every function returns int32 status (BDR_OK == 0);
 *     bdr_last_error() returns a thread-local message for the last failure.
 *   - handles are opaque, heap allocated, NOT thread-safe but movable between threads
 *     (Rust `Send`, not `Sync`), one HIP stream per handle, hipSetDevice on every entry.
 *   - host pointers passed in are copied before the call returns unless stated otherwise.
 *   - there is no CPU fallback: without a HIP device every constructor fails with
 *     BDR_ERR_NO_DEVICE.
 */
#ifndef BORDER_AMD_H
#define BORDER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDR_API __attribute__((visibility("default")))

enum {
    BDR_OK = 0,
    BDR_ERR_INVALID = 1,    /* bad argument / config (anyhow::Error in the shim)     */
    BDR_ERR_NO_DEVICE = 2,  /* no HIP device / extension cannot run                  */
    BDR_ERR_HIP = 3,        /* a HIP runtime call failed                             */
    BDR_ERR_EMPTY = 4,      /* batch() on an empty buffer (the reference panics)     */
    BDR_ERR_IO = 5,         /* save/load failure                                     */
    BDR_ERR_COMM = 6        /* RCCL failure                                          */
};

typedef struct bdr_replay bdr_replay;
typedef struct bdr_agent bdr_agent;
typedef struct bdr_comm bdr_comm;

BDR_API const char* bdr_last_error(void);
/* 1 when the last failure on this thread reports a device-side condition of an EARLIER asynchronous step (an action index outside
 * [0, n_actions) that reached a TD step, a cross-queue wait that timed out, a non-finite priority): the call that returned it did NOT
 * fail on its own arguments, the condition has been cleared and the agent's state is that of the last good update.  Agent::opt returns
 * () in the reference (border-core/src/base/agent.rs:24-136): a caller of bdr_agent_opt may log such a report and go on, and must
 * treat every other failure (BDR_ERR_EMPTY, a buffer that does not match the agent, a HIP error of the call itself) as the reference's
 * panic.  0 otherwise.
 * 2 is the same kind of report raised by the step THE FAILING CALL ITSELF ran: bdr_agent_opt_with_record / _with_scalars enqueue the
 * step, synchronise and read the error words once more - when those flag this very step (its own TD kernel met an out-of-range action,
 * its own tree update a NaN priority) the step HAS run (n_opts and the optimizer step advanced once) and the record it produced is
 * complete in the caller's buffers (rec / out, *n_out).  A caller logs the report and keeps the record; it must NOT run the call
 * again to "retry" - that would be a second optimisation step.  After 1 the step of the failing call was never enqueued. */
BDR_API int32_t bdr_last_error_is_deferred(void);
BDR_API int32_t bdr_device_count(int32_t* count);
BDR_API const char* bdr_version(void);

/* ------------------------------------------------------------------------------------------
 * SimpleReplayBuffer  (border-core/src/generic_replay_buffer/base.rs:86-123)
 * ---------------------------------------------------------------------------------------- */

/* SimpleReplayBufferConfig (generic_replay_buffer/config.rs:185-210): capacity, seed; per_config = Some(..) is
 * bdr_replay_enable_per on the empty buffer (below). */
typedef struct {
    uint64_t capacity;
    uint64_t seed;
    uint64_t obs_row_bytes; /* bytes of one observation row (Atari: 4*1*84*84 u8 = 28224)  */
    uint64_t act_row_bytes; /* bytes of one action row (discrete: 8, one i64)              */
    int32_t device;         /* HIP device ordinal                                          */
    int32_t frame_stack;    /* 0: rows stored as given.  k > 0 (Atari: 4): SINGLE-FRAME store - an observation is k frames of
                             * obs_row_bytes / k bytes, newest first (border-atari-env/src/env.rs:197-209 stack_frame), and every
                             * distinct frame is stored once: obs_t / next_obs_t share k-1 frames and next_obs_t == obs_t+1 inside
                             * an episode, so a transition costs ONE new frame instead of 2k (8x less HBM for k = 4).  push()
                             * still takes the stacked rows and finds the sharing by comparing bytes (any input is stored
                             * exactly; what cannot be shared is stored in full), batch() rebuilds the stacks - indices, pushes
                             * and batches are identical to the plain ring's. */
    uint64_t frame_capacity;/* frames in the store (0: capacity + capacity / 4 + 64).  A push that would overwrite a frame a
                             * live transition still references fails with BDR_ERR_INVALID (raise frame_capacity). */
    int32_t index_rng;      /* BDR_RNG_STDRNG (0, default): the index stream of the reference's StdRng::seed_from_u64(seed)
                             * (base.rs:353, 386) bit for bit.  BDR_RNG_XOSHIRO256PP (1): north_star's "on-device xoshiro index
                             * generator" - one xoshiro256++ generator per batch lane in HBM (lane j seeded with outputs 4j .. 4j+3
                             * of SplitMix64(seed)), lane j draws sample j of every batch: ix = (next_u64() >> 32) % size.  NOT the
                             * reference's stream (no parity claim: a device-native alternative for hosts that do not need one);
                             * deterministic in (seed, the sequence of batch sizes).  Uniform sampling only. */
    int32_t reserved;
} bdr_replay_config;
#define BDR_RNG_STDRNG 0
#define BDR_RNG_XOSHIRO256PP 1

/* ReplayBufferBase::build (base.rs:336-356).  The ring lives in HBM as one fused record per
 * transition: [obs | next_obs | act | reward f32 | is_terminated i8 | is_truncated i8 | pad]. */
BDR_API int32_t bdr_replay_create(const bdr_replay_config* cfg, bdr_replay** out);
BDR_API int32_t bdr_replay_destroy(bdr_replay* r);

/* ExperienceBufferBase::push (base.rs:295-316): n transitions, rows written at
 * (i+k) % capacity, i = (i+n) % capacity, size = min(size+n, capacity). Host pointers.  The call returns when the
 * caller's buffers are free (the rows sit in the buffer's pinned staging area); the copy into the ring completes on
 * the buffer's stream, in order before every later batch / read of this buffer. */
BDR_API int32_t bdr_replay_push(bdr_replay* r, uint64_t n, const void* obs, const void* act,
                                const void* next_obs, const float* reward,
                                const int8_t* is_terminated, const int8_t* is_truncated);

/* The same push for transitions whose observation rows already live in HBM - the frame stacks of a bdr_atari_prep (obs = its
 * stacks before the step, bdr_atari_prep_device_prev_stacks; next_obs = after it, bdr_atari_prep_device_stacks), or any device
 * rows `*_stride` bytes apart: trainer/sampler.rs:99-144 + border-atari-env/src/env.rs:197-209,312-324 without the HBM -> host ->
 * HBM round trip of the stacks.  act / reward / flags are host arrays (they come from the host side: exploration, the emulator).
 * Ring rows, cursor, size and PER priorities are exactly those of bdr_replay_push on the same bytes.  Not for frame_stack > 0
 * buffers (their push compares host rows): BDR_ERR_INVALID.  Returns when the rows have been copied. */
BDR_API int32_t bdr_replay_push_device(bdr_replay* r, uint64_t n, const void* obs_dev, uint64_t obs_stride, const void* act,
                                       const void* next_obs_dev, uint64_t next_obs_stride, const float* reward,
                                       const int8_t* is_terminated, const int8_t* is_truncated);

/* ExperienceBufferBase::len (base.rs:318-320) and the write cursor `i`. */
BDR_API int32_t bdr_replay_len(const bdr_replay* r, uint64_t* len);
BDR_API int32_t bdr_replay_head(const bdr_replay* r, uint64_t* head);

/* Single-frame store only: frames allocated since creation (a continuing episode costs one per transition) and the store's
 * capacity in frames. */
BDR_API int32_t bdr_replay_frames_used(const bdr_replay* r, uint64_t* allocated, uint64_t* capacity);

/* The index draw of ReplayBufferBase::batch (base.rs:384-390):
 * ixs[k] = (StdRng::next_u32() as usize) % size, generated ON DEVICE by the ChaCha12 kernel
 * and copied back; advances the RNG exactly as one batch(n) call does. */
BDR_API int32_t bdr_replay_sample_indices(bdr_replay* r, uint64_t n, uint64_t* ixs_out);

/* ReplayBufferBase::batch (base.rs:376-402): draws indices and gathers the batch into
 * device-resident batch buffers owned by the replay handle (valid until the next batch()).
 * Any *_out host pointer may be NULL; non-NULL ones receive a copy (synchronises).
 * GenericTransitionBatch fields (generic_replay_buffer/batch.rs:89-162): obs, act, next_obs,
 * reward, is_terminated, is_truncated, ix_sample; weight is None (uniform sampling). */
BDR_API int32_t bdr_replay_batch(bdr_replay* r, uint64_t n, uint64_t* ixs_out, void* obs_out,
                                 void* act_out, void* next_obs_out, float* reward_out,
                                 int8_t* is_terminated_out, int8_t* is_truncated_out);

/* Device pointers of the last gathered batch (for callers that keep the batch on device). */
typedef struct {
    uint64_t n;
    const void* obs;      /* [n][obs_row_bytes] */
    const void* next_obs; /* [n][obs_row_bytes] */
    const void* act;      /* [n][act_row_bytes] */
    const float* reward;  /* [n] */
    const int8_t* is_terminated;
    const int8_t* is_truncated;
    const uint64_t* ixs;
    const float* weight;  /* [n] importance weights when PER is enabled, else NULL */
} bdr_device_batch;
BDR_API int32_t bdr_replay_last_batch(const bdr_replay* r, bdr_device_batch* out);

/* Benchmark helper (SURVEY.md section 8(d) synthetic inputs): fills transitions [0,n) on the
 * device from a counter-based generator (seed, transition index) and sets size=min(n,capacity),
 * i = n % capacity.  kind 0: Atari-like (u8 frames uniform 0..255, act uniform i64 in
 * [0,n_actions), reward in {-1,0,1} with P=(.05,.9,.05), P(term)=.005, trunc 0);
 * kind 1: f32 rows ~ N(0,1) approx (obs), act f32 ~ U(-1,1) (n_actions==0) or i64 uniform. */
BDR_API int32_t bdr_replay_fill_synthetic(bdr_replay* r, uint64_t n, uint64_t seed, int32_t kind,
                                          int32_t n_actions);
/* Test helper: copy ring rows [first, first+n) back to the host (any pointer may be NULL). */
BDR_API int32_t bdr_replay_read_rows(bdr_replay* r, uint64_t first, uint64_t n, void* obs, void* act,
                                     void* next_obs, float* reward, int8_t* term, int8_t* trunc);

/* ------------------------------------------------------------------------------------------
 * Prioritized experience replay  (SimpleReplayBufferConfig::per_config, config.rs:45-83;
 * generic_replay_buffer/base/sum_tree.rs; base/iw_scheduler.rs)
 * ---------------------------------------------------------------------------------------- */
enum { BDR_PER_NORMALIZE_ALL = 0, BDR_PER_NORMALIZE_BATCH = 1 };   /* WeightNormalizer, sum_tree.rs:13-18 */
typedef struct {
    float alpha;            /* 0.6   (config.rs:76) */
    float beta_0;           /* 0.4 */
    float beta_final;       /* 1.0 */
    uint64_t n_opts_final;  /* 500_000 */
    int32_t normalize;      /* BDR_PER_NORMALIZE_ALL */
    int32_t reserved;
} bdr_per_config;
BDR_API void bdr_per_config_default(bdr_per_config* c);

/* SimpleReplayBuffer::build with per_config: Some(..) (base.rs:336-356).  Must be called on an empty
 * buffer.  From then on push() gives new rows the current maximum priority (set_priority, :227-235),
 * batch() samples through the sum tree and carries importance weights (:377-383), and agents that
 * consume weights (DQN, dqn/base.rs:123-145) call update_priority with their TD errors. */
BDR_API int32_t bdr_replay_enable_per(bdr_replay* r, const bdr_per_config* c);

/* ReplayBufferBase::update_priority (base.rs:413-426): sum_tree.update(ix, td_err) in order, then the
 * importance-weight schedule advances by one optimisation step.  Host arrays. */
BDR_API int32_t bdr_replay_update_priority(bdr_replay* r, uint64_t n, const uint64_t* ixs, const float* td_errs);

/* `weight` of the last batch() (Some(ws), base.rs:382): n floats.  Error when PER is not enabled. */
BDR_API int32_t bdr_replay_batch_weights(bdr_replay* r, uint64_t n, float* w_out);

/* Introspection for parity tests: scheduler state, SumTree::{total,max}, the raw arrays
 * (what: 0 sum tree [2*capacity-1] in the reference's layout, 1 {min over [0,n_samples), max} as two floats), SumTree::get. */
typedef struct { uint64_t n_samples, n_opts; float beta, total, max_p, min_p; } bdr_per_info;
BDR_API int32_t bdr_replay_per_info(bdr_replay* r, bdr_per_info* out);
BDR_API int32_t bdr_replay_per_read(bdr_replay* r, int32_t what, float* out, uint64_t n);
BDR_API int32_t bdr_replay_per_get(bdr_replay* r, float s, uint64_t* ix);

/* ------------------------------------------------------------------------------------------
 * DQN agent  (border-tch-agent/src/dqn/base.rs, dqn/config.rs:26-48, dqn/model/base.rs)
 * ---------------------------------------------------------------------------------------- */

enum { BDR_NET_ATARI_CNN = 0, BDR_NET_MLP = 1 };
enum { BDR_LOSS_MSE = 0, BDR_LOSS_SMOOTH_L1 = 1 };        /* util.rs:17-23 CriticLoss      */
enum { BDR_OPT_ADAM = 0, BDR_OPT_ADAMW = 1 };             /* opt.rs:13-28 OptimizerConfig  */
#define BDR_MAX_UNITS 8

/* Which products the large matrix layers of an agent compute.  NOT a reference field: the reference has one arithmetic (f32
 * products, f32 accumulation in ATen's summation order).  Both settings accumulate in f32 and sit inside the 1e-4 parity bar:
 *   BDR_ARITH_BF16X3_6   (default) every f32 operand of conv2 / conv3 forward (DQN) and of the merge / cosine-embedding layers
 *                        (IQN at C4-sized shapes) is split, round-to-nearest, into three bf16 terms (a = a0 + a1 + a2 exactly up to
 *                        the last bit) and six of the nine partial products run on the bf16 MFMA; the three dropped products are
 *                        <= 2^-24 relative per product, zero-mean.  Every other layer, every gradient and the n <= 8 acting
 *                        kernels compute exact f32 products.
 *   BDR_ARITH_F32_EXACT  every product is an f32 x f32 product on the FP32 MFMA (v_mfma_f32_32x32x2_f32).
 * The environment variables BDR_DQN_F32_EXACT / BDR_IQN_F32_EXACT (=1 exact, =0 split) override the field for A/B runs only. */
enum { BDR_ARITH_BF16X3_6 = 0, BDR_ARITH_F32_EXACT = 1 };

/* OptimizerConfig's AdamW variant (opt.rs:20-27, 38-55) beside a model's `lr`: opt_kind BDR_OPT_ADAM ignores every other field
 * (tch nn::Adam::default(): .9, .999, 1e-8, wd 0); amsgrad keeps a max_exp_avg_sq arena (parameter model +400 / arena 5). */
typedef struct {
    int32_t opt_kind; /* BDR_OPT_* */
    int32_t amsgrad;
    double beta1, beta2, weight_decay, eps;
} bdr_adamw_config;

/* AtariCnnConfig (cnn/config.rs:13-18) / MlpConfig (mlp/config.rs:7-12) */
typedef struct {
    int32_t kind;    /* BDR_NET_* */
    int32_t n_stack; /* cnn */
    int32_t in_dim;  /* mlp */
    int32_t n_units; /* mlp */
    int32_t units[BDR_MAX_UNITS];
    int32_t out_dim; /* number of actions */
    int32_t activation_out;
} bdr_net_config;

/* DqnConfig (dqn/config.rs:26-48; defaults :82-102) + DqnModelConfig.opt_config. */
typedef struct {
    bdr_net_config net;
    int32_t opt_kind; /* BDR_OPT_ADAM: lr only (tch nn::Adam::default(): .9,.999,1e-8,wd 0) */
    double lr;
    double beta1, beta2, weight_decay, eps; /* AdamW variant only */
    int32_t amsgrad;  /* AdamW{amsgrad} (opt.rs:20-27): denominator from the running max of exp_avg_sq (arena 5) */
    uint64_t soft_update_interval;
    uint64_t n_updates_per_opt;
    uint64_t batch_size;
    double discount_factor;
    double tau;
    int32_t train;
    int32_t double_dqn;
    int32_t critic_loss; /* BDR_LOSS_* */
    int32_t has_clip_td_err;
    double clip_td_err_min, clip_td_err_max;
    int32_t record_verbose_level;
    int32_t device; /* HIP device ordinal ("No device is given for DQN agent": dqn/base.rs:256) */
    uint64_t param_seed; /* seed of the library's own uniform(+-1/sqrt(fan_in)) initialiser */
    int32_t arithmetic;  /* BDR_ARITH_* (AtariCnn only; an Mlp Q-network always computes exact f32 products) */
    int32_t reserved;
} bdr_dqn_config;

BDR_API void bdr_dqn_config_default(bdr_dqn_config* cfg); /* dqn/config.rs:82-102 */

/* Configurable::build (dqn/base.rs:252-286). */
BDR_API int32_t bdr_dqn_create(const bdr_dqn_config* cfg, bdr_agent** out);
BDR_API int32_t bdr_agent_destroy(bdr_agent* a);

/* Agent::train / eval / is_train (dqn/base.rs:289-299). */
BDR_API int32_t bdr_agent_set_train(bdr_agent* a, int32_t train);
BDR_API int32_t bdr_agent_is_train(const bdr_agent* a, int32_t* out);

/* Agent::opt (border-core/src/base/agent.rs:62-64 -> dqn/base.rs:301-309 -> opt_ :182-200):
 * n_updates_per_opt x update_critic, soft-update bookkeeping, n_opts += 1.
 * Enqueues on the agent's stream and returns WITHOUT synchronising. */
BDR_API int32_t bdr_agent_opt(bdr_agent* a, bdr_replay* buffer);

/* Agent::opt_with_record (dqn/base.rs:311-343): same step, then synchronises and returns
 * the Record scalars.  keys: "loss" (always); verbose>=2 adds pred_mean, reward_mean,
 * tgt_mean, tgt_minus_pred_mean. */
typedef struct {
    float loss;
    float pred_mean, reward_mean, tgt_mean, tgt_minus_pred_mean;
    int32_t has_verbose;
} bdr_dqn_record;
BDR_API int32_t bdr_agent_opt_with_record(bdr_agent* a, bdr_replay* buffer, bdr_dqn_record* rec);

/* Agent::opt_with_record for any agent kind: the Record scalars in the agent's documented order
 * (DQN: loss[, pred_mean, reward_mean, tgt_mean, tgt_minus_pred_mean]; IQN: loss_critic;
 * SAC: loss_critic, loss_actor, ent_coef).  Synchronises. */
BDR_API int32_t bdr_agent_opt_with_scalars(bdr_agent* a, bdr_replay* buffer, float* out, int32_t cap, int32_t* n_out);

/* Names of the scalars bdr_agent_opt_with_scalars returns, '\n'-separated, in order (the keys of the reference's Record):
 *   DQN  "loss"; record_verbose_level >= 2 adds pred_mean, reward_mean, tgt_mean, tgt_minus_pred_mean (dqn/base.rs:107-121),
 *        then qnet.param_stats() - "<var>_mean", "<var>_std" (population) for c1.weight ... l2.bias / mlp.ln{i}.* in the
 *        variables' order (util.rs:64-80) - and "ratio_best_act" = n_samples_best_act / n_samples_act, which resets both
 *        counters (dqn/base.rs:316-342);
 *   IQN  "loss_critic" (iqn/base.rs:190);   SAC  "loss_critic", "loss_actor", "ent_coef" (sac/base.rs:187-196). */
BDR_API int32_t bdr_agent_record_keys(bdr_agent* a, char* names_out, uint64_t names_cap, int32_t* n_keys);

/* Test helper: n draws of the agent's own device noise stream copied to the host - SAC: the N(0,1) draws of action_logp
 * (sac/base.rs:76 uses torch's global generator), IQN: the U[0,1) percent points of IqnSample::Uniform* (iqn/model/base.rs:365-368).
 * Counter-based (seed, running counter): advances the stream exactly as an update consuming n draws does. */
BDR_API int32_t bdr_agent_draw_noise(bdr_agent* a, uint64_t n, float* out);

/* One update_critic on a caller-supplied host minibatch (parity tests: "fixed minibatch").
 * act: int64 [n]; obs/next_obs rows as in the replay buffer. */
BDR_API int32_t bdr_dqn_update_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act,
                                        const void* next_obs, const float* reward,
                                        const int8_t* is_terminated, bdr_dqn_record* rec);

/* The `if let Some(ws) = weight` branch of update_critic (dqn/base.rs:123-145) on a caller-supplied minibatch:
 * weight[n] importance weights (NULL = the unweighted branch); td_errs_out[n] (optional) receives
 * |pred - tgt| (clipped by clip_td_err), the values the reference hands to update_priority. */
BDR_API int32_t bdr_dqn_update_on_batch_weighted(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act,
                                                 const void* next_obs, const float* reward, const int8_t* is_terminated,
                                                 const float* weight, float* td_errs_out, bdr_dqn_record* rec);

/* Split step (synchronous data-parallel training): bdr_dqn_grads_on_batch = update_critic up to
 * `loss.backward()` (dqn/base.rs:60-150, opt.rs:74-83 without `step`) on a host minibatch - gradients land in arena 4
 * (bdr_agent_get_params / set_params), parameters, Adam moments and counters are untouched; bdr_agent_apply_grads = the
 * optimizer step on whatever arena 4 holds, then opt_'s bookkeeping (soft-update counter, n_opts).  N ranks that each take the
 * gradient of B/N rows, average the arenas and apply take exactly the step one rank takes on the B rows. */
BDR_API int32_t bdr_dqn_grads_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act, const void* next_obs,
                                       const float* reward, const int8_t* is_terminated, bdr_dqn_record* rec);
BDR_API int32_t bdr_agent_apply_grads(bdr_agent* a);

/* Q(obs) for n observations -> q_out[n][A] and argmax actions (either pointer may be NULL).
 * DQN: qnet.forward (dqn/base.rs:213); IQN: quantile average (iqn/base.rs:209-215). */
BDR_API int32_t bdr_agent_qvalues(bdr_agent* a, uint64_t n, const void* obs, float* q_out,
                                  int64_t* argmax_out);

/* DqnExplorer / IqnExplorer (dqn/explorer.rs:8-15, iqn/explorer.rs:9-16) with its mutable state.
 * The reference draws from fastrand's global unseeded generator; this library draws from a seeded
 * ChaCha12 stream (StdRng::seed_from_u64(seed)), so action sequences are reproducible:
 *   eps-greedy call:  eps = max(eps_start - (eps_start-eps_final)/final_step * n_calls, eps_final);
 *                     coin = f64 (one u64 of the stream, top 52 bits); n_calls += 1;
 *                     coin < eps ? n_procs x below(A) (one draw each, row order) : argmax per row
 *   softmax call:     per row one f64 u, action = first k with cumsum(softmax(q))[k] > u
 *   eval (DQN only):  f32 (one u32, top 23 bits) < 0.01 ? one below(A) for every row : argmax   */
enum { BDR_EXPLORER_SOFTMAX = 0, BDR_EXPLORER_EPS_GREEDY = 1 };
typedef struct {
    int32_t kind;          /* BDR_EXPLORER_* */
    double eps_start;      /* 1.0  (explorer.rs:47) */
    double eps_final;      /* 0.02 */
    uint64_t final_step;   /* 100_000; dqn_atari: 1_000_000 */
    uint64_t n_calls;      /* the reference's `n_opts` field of EpsilonGreedy: action() calls so far */
    uint64_t seed;         /* seed of the exploration stream (set_explorer rewinds the stream) */
} bdr_explorer_config;
BDR_API void bdr_explorer_config_default(bdr_explorer_config* e, int32_t kind);
BDR_API int32_t bdr_agent_set_explorer(bdr_agent* a, const bdr_explorer_config* e);
BDR_API int32_t bdr_agent_get_explorer(const bdr_agent* a, bdr_explorer_config* e);

/* Policy::sample (dqn/base.rs:211-242, iqn/base.rs:204-228) for n_procs observations (host rows as in the
 * replay buffer): device forward, exploration as configured, act_out[n_procs] (int64).
 * train: n_samples_act += 1 (and n_samples_best_act += 1 when every row took its greedy action, the
 * reference's `record_verbose_level >= 2` bookkeeping).  info may be NULL.
 * ONE network, two kernel families (AtariCnn DQN / IQN): calls with n_procs <= 8 run the acting kernels (csrc/act_small.hpp: exact f32
 * products from the f32 weights, k-sliced 32 x 32 tiles), larger calls the training forward (conv2 / conv3 with the agent's
 * `arithmetic`).  The two evaluate the same function to f32 round-off - Q-values within 5e-6 of the largest |Q| with BDR_ARITH_BF16X3_6,
 * 1e-6 with BDR_ARITH_F32_EXACT (tests/test_gpu_sample.py) - but not bit for bit: the greedy action of a row is the same on both
 * whenever its two leading Q-values differ by more than 2e-5 of the largest |Q|; below that distance it may depend on how many
 * environments share the call (8 vs 9).  Exact ties (equal rows of the output layer) resolve to the first maximum on both.  The
 * reference has the same property across batch sizes (ATen picks its GEMM blocking by shape); a caller that needs one family for
 * every n sets BDR_NO_ACT_SMALL=1 (training forward always, ~2x the one-observation latency). */
typedef struct {
    double eps;            /* eps used by this call (eps-greedy), else 0 */
    int32_t is_random;     /* the call took the random branch */
    uint64_t n_samples_act, n_samples_best_act;
} bdr_sample_info;
BDR_API int32_t bdr_agent_sample(bdr_agent* a, uint64_t n_procs, const void* obs, int64_t* act_out,
                                 bdr_sample_info* info);
/* Policy::sample for observations that are already on the device (trainer/sampler.rs:99-144 with the
 * observation of border-atari-env/src/env.rs:197-209 kept in HBM): row i at obs_dev + i * row_stride bytes, e.g.
 * bdr_atari_prep_device_stacks with row_stride = 4*84*84.  Forward, exploration stream, counters and results are those of
 * bdr_agent_sample on the same bytes; contiguous rows are read in place.  The rows must be complete when the call is made and
 * stay untouched until it returns.  bdr_agent_qvalues_device: the action values themselves (bdr_agent_qvalues). */
BDR_API int32_t bdr_agent_sample_device(bdr_agent* a, uint64_t n_procs, const void* obs_dev, uint64_t row_stride, int64_t* act_out,
                                        bdr_sample_info* info);
BDR_API int32_t bdr_agent_qvalues_device(bdr_agent* a, uint64_t n, const void* obs_dev, uint64_t row_stride, float* q_out,
                                         int64_t* argmax_out);

/* Block until everything enqueued on the agent's stream has finished.  Device-side error flags raised since the last check
 * (an action index outside [0, n_actions) in a TD step, a NaN priority in update_priority, a timed-out cross-queue gate) are
 * reported here - and by every other entry point that synchronises (opt_with_record / _with_scalars, get_params, save_params,
 * sample, update_on_batch); bdr_agent_opt polls them without synchronising every 256 calls. */
BDR_API int32_t bdr_agent_sync(bdr_agent* a);
BDR_API int32_t bdr_agent_n_opts(const bdr_agent* a, uint64_t* n);

/* SyncModel::model_info / sync_model (border-async-trainer/src/sync_model.rs:2-13,
 * dqn/base.rs:377-402) and checkpoint access.  Parameters cross the boundary in the
 * reference's variable order and layouts (c1.weight OIHW, c1.bias, ... l2.bias / mlp.ln{i}.*).
 * which: 0 = qnet, 1 = qnet_tgt, 2 = Adam exp_avg, 3 = Adam exp_avg_sq, 4 = last gradient, 5 = AdamW amsgrad max_exp_avg_sq. */
BDR_API int32_t bdr_agent_param_count(const bdr_agent* a, uint64_t* n);
BDR_API int32_t bdr_agent_param_count_of(bdr_agent* a, int32_t which, uint64_t* n); /* per-model counts (SAC) */
BDR_API int32_t bdr_agent_get_params(bdr_agent* a, int32_t which, float* out, uint64_t n);
BDR_API int32_t bdr_agent_set_params(bdr_agent* a, int32_t which, const float* in, uint64_t n);

/* Device address of a flat parameter arena (internal kernel layout, identical on every rank) so a
 * host that already owns a communicator (e.g. torch.distributed) can reduce it in place.
 * Cost: from this call on the library cannot know when arena `which` is written, so every copy it derives from those parameters
 * (AtariCnn DQN: the bf16 planes of W2 / W3 the split-operand forward reads) is rebuilt before EVERY forward of that parameter set
 * (one ~4 us launch per forward) - until bdr_agent_arena_release(a, which) hands the arena back. */
BDR_API int32_t bdr_agent_arena_device_ptr(bdr_agent* a, int32_t which, void** ptr, uint64_t* n_floats);
/* "I no longer write through the pointer bdr_agent_arena_device_ptr gave me": derived copies are rebuilt once more and trusted again. */
BDR_API int32_t bdr_agent_arena_release(bdr_agent* a, int32_t which);

/* Agent::save_params / load_params (dqn/base.rs:345-371; iqn/base.rs:303-317; sac/base.rs:313-345): writes / reads
 * `qnet.pt.tch`, `qnet_tgt.pt.tch` (IQN: iqn, iqn_tgt; SAC: pi, qnet_{i}, qnet_tgt_{i}, ent_coef) under dir - the
 * reference's file names.  The container follows the file name exactly as tch's VarStore::{save,load} do:
 *   BDR_CKPT_TCH          "<stem>.pt.tch": the libtorch named-tensor archive (TorchScript module zip) that tch writes
 *                         through torch-sys at_save_multi and reads with torch::jit::load (default, = the reference);
 *   BDR_CKPT_SAFETENSORS  "<stem>.safetensors".
 * Both hold the reference's variable names in the reference's layouts (OIHW conv weights, [out,in] linear weights).
 * Loading prefers the configured container and falls back to the other one when only that file exists; it requires
 * every variable of the model with its shape and dtype float32; extra entries are ignored. */
#define BDR_CKPT_TCH 0
#define BDR_CKPT_SAFETENSORS 1
BDR_API int32_t bdr_agent_set_checkpoint_format(bdr_agent* a, int32_t format);
BDR_API int32_t bdr_agent_save_params(bdr_agent* a, const char* dir);
BDR_API int32_t bdr_agent_load_params(bdr_agent* a, const char* dir);

/* ---- host-side Trainer loops (border-core/src/trainer.rs) -------------------------------------------------------------
 * The reference's Trainer is compiled Rust; this is its compiled counterpart above the C ABI (csrc/trainer.hip, host code).
 * Agent, buffer and environment are reached through function tables, so a border maintainer can drive the library's
 * handles (bdr_trainer_ops_default) or their own objects, and the loop rules are testable without a GPU.
 *   bdr_trainer_train          Trainer::train (trainer.rs:267-327): Sampler::sample_and_push (trainer/sampler.rs:99-144, with
 *                              SimpleStepProcessor, generic_replay_buffer/step_proc.rs:62-137), then train_step (:197-228):
 *                              no opt while env_steps < warmup_period or env_steps % opt_interval != 0; opt_with_record when
 *                              (opt_steps + 1) % record_agent_info_interval == 0; the timer wraps opt*() only; stop at
 *                              opt_steps == max_opts
 *   bdr_trainer_train_offline  Trainer::train_offline (:330-384): warmup_period = 0, opt_interval = 1, no sampling
 * Recorder / evaluator sinks are out of scope; `observer` receives what the reference would store. */
typedef struct bdr_env_vtable {             /* single-process Env (border-core/src/base/env.rs:45-181) */
    void* ctx;
    int32_t (*reset)(void* ctx, void* obs_out);                                  /* Env::reset(None) */
    /* Env::step_with_reset(&act) (env.rs:137-161): writes obs, reward, flags; when the step is done also init_obs */
    int32_t (*step_with_reset)(void* ctx, const void* act, void* obs_out, float* reward, int8_t* is_terminated,
                               int8_t* is_truncated, void* init_obs_out);
    /* Device-resident observations (zero-initialised tables keep the host convention): with obs_on_device != 0
     * obs_out / init_obs_out are DEVICE buffers of obs_row_bytes on GPU `device` - the environment keeps its observation in HBM
     * (a bdr_atari_prep: bdr_atari_prep_copy_stack writes an environment's stack there) - and the loops act and push through the
     * *_device entries of their function tables: the observation never visits the host. */
    int32_t obs_on_device;
    int32_t device;
} bdr_env_vtable;

typedef struct bdr_trainer_ops {
    void* agent;
    void* buffer;
    int32_t (*agent_set_train)(void* agent, int32_t train);                                       /* Agent::train */
    int32_t (*agent_sample)(void* agent, uint64_t n_procs, const void* obs, void* act_out);       /* Policy::sample */
    int32_t (*agent_opt)(void* agent, void* buffer);                                              /* Agent::opt */
    int32_t (*agent_opt_with_record)(void* agent, void* buffer, float* scalars, int32_t cap, int32_t* n_scalars);
    int32_t (*buffer_push)(void* buffer, uint64_t n, const void* obs, const void* act, const void* next_obs,
                           const float* reward, const int8_t* is_terminated, const int8_t* is_truncated);
    /* used instead of agent_sample / buffer_push when the environment's observations are device-resident (bdr_env_vtable::obs_on_device):
     * bdr_agent_sample_device / bdr_replay_push_device by default; act_out, act, reward and the flags stay host memory */
    int32_t (*agent_sample_device)(void* agent, uint64_t n_procs, const void* obs_dev, uint64_t row_stride, void* act_out);
    int32_t (*buffer_push_device)(void* buffer, uint64_t n, const void* obs_dev, uint64_t obs_stride, const void* act,
                                  const void* next_obs_dev, uint64_t next_obs_stride, const float* reward,
                                  const int8_t* is_terminated, const int8_t* is_truncated);
} bdr_trainer_ops;

typedef struct bdr_trainer_config {         /* trainer/config.rs:30-87; an interval of 0 means "never" */
    uint64_t max_opts;
    uint64_t opt_interval;
    uint64_t warmup_period;
    uint64_t record_agent_info_interval;
    uint64_t record_compute_cost_interval;
    uint64_t obs_row_bytes;                 /* row sizes of the environment's observation / action (online loop) */
    uint64_t act_row_bytes;
} bdr_trainer_config;

typedef struct bdr_trainer_stats {
    uint64_t env_steps, opt_steps, n_records, n_episodes;
    double opt_seconds, sample_seconds;     /* timer_for_opt_steps / timer_for_samples summed over the run */
} bdr_trainer_stats;

#define BDR_TRAINER_EVENT_SKIP 0            /* iteration without an opt step */
#define BDR_TRAINER_EVENT_OPT 1             /* Agent::opt */
#define BDR_TRAINER_EVENT_OPT_RECORD 2      /* Agent::opt_with_record: scalars = the agent's Record */
#define BDR_TRAINER_EVENT_COST 3            /* scalars = {average_opt_time, average_sample_time} in ms (trainer.rs:164-181) */
typedef void (*bdr_trainer_observer)(void* ctx, uint64_t env_steps, uint64_t opt_steps, int32_t event, const float* scalars,
                                     int32_t n_scalars);

BDR_API void bdr_trainer_config_default(bdr_trainer_config* c);
BDR_API void bdr_trainer_ops_default(bdr_trainer_ops* ops, bdr_agent* agent, bdr_replay* buffer);
BDR_API int32_t bdr_trainer_train(const bdr_trainer_config* c, const bdr_trainer_ops* ops, const bdr_env_vtable* env,
                                  bdr_trainer_observer observer, void* observer_ctx, bdr_trainer_stats* out);
BDR_API int32_t bdr_trainer_train_offline(const bdr_trainer_config* c, const bdr_trainer_ops* ops, bdr_trainer_observer observer,
                                          void* observer_ctx, bdr_trainer_stats* out);

/* ---- border-async-trainer (border-async-trainer/src) -----------------------------------------------------------------------
 * The reference runs one learner thread (AsyncTrainer::train, async_trainer/base.rs:299-388) and n actor threads (Actor::run,
 * actor/base.rs:120-178) in one process: actors sample with their own agent + env, buffer n_buffer transitions in a
 * ReplayBufferProxy (replay_buffer_proxy.rs:52-72) and try_send them as one PushedItemMessage; the learner drains the channel
 * into its replay buffer (update_replay_buffer, :275-284), does not optimise until buffer.len() >= warmup_period (:210,
 * 328-335), and every sync_interval opt steps sends (n_opts, model_info) back (sync, :268-272); an actor adopts a model that is
 * newer than the one it has (actor/base.rs:104-118).
 * Here: csrc/async_trainer.hip, compiled, behind function tables like the Trainer above.  One rank per GPU = one learner + its
 * actors + its local replay shard; `exchange` (optional) runs at every sync point before the local publish and is where the
 * ranks average their learners over RCCL (bdr_agent_allreduce_params).  The model channel is a device-resident mailbox:
 * publish = one device-to-device copy on the learner's stream, sync = one copy on the actor's stream, ordered by events. */
typedef struct bdr_model_mailbox bdr_model_mailbox;
BDR_API int32_t bdr_model_mailbox_create(int32_t device, uint64_t n_floats, uint32_t n_readers, bdr_model_mailbox** out);
BDR_API int32_t bdr_model_mailbox_destroy(bdr_model_mailbox* m);
/* SyncModel::model_info + send (dqn/base.rs:377-389, async_trainer/base.rs:268-272): arena `which` (bdr_agent_arena_device_ptr)
 * of the learner -> mailbox, tagged n_opts.  Asynchronous on the agent's stream. */
BDR_API int32_t bdr_agent_publish_model(bdr_agent* a, int32_t which, bdr_model_mailbox* m, uint64_t n_opts);
/* Actor::sync_model (actor/base.rs:104-118): copies the mailbox into the agent iff mailbox.n_opts > *n_opts_inout (or `first`:
 * sync_model_first, :98-102), then *n_opts_inout = mailbox.n_opts, *updated = 1.  reader: the actor's index (< n_readers). */
BDR_API int32_t bdr_agent_sync_model_from(bdr_agent* a, int32_t which, bdr_model_mailbox* m, uint32_t reader, int32_t first,
                                          uint64_t* n_opts_inout, int32_t* updated);

typedef struct bdr_async_trainer_config {   /* async_trainer/config.rs:13-36 (defaults :101-112) + ActorManagerConfig::n_buffer */
    uint64_t max_opts;
    uint64_t warmup_period;                 /* transitions in the learner's buffer before the first opt */
    uint64_t sync_interval;                 /* opt steps between model syncs (default 100; dqn_atari_async_tch: 1) */
    uint64_t record_agent_info_interval;    /* 0 = never */
    uint64_t record_compute_cost_interval;  /* 0 = never */
    uint64_t n_buffer;                      /* ReplayBufferProxyConfig::n_buffer: transitions per message (100) */
    uint64_t channel_capacity;              /* bounded(1000), actor_manager/base.rs:140 */
    uint64_t warmup_sleep_ms;               /* the 100 ms pause between warm-up and training (async_trainer/base.rs:331) */
    uint64_t obs_row_bytes, act_row_bytes;
} bdr_async_trainer_config;

typedef struct bdr_learner_ops {
    bdr_trainer_ops t;                      /* agent, buffer, set_train, opt, opt_with_record, buffer_push */
    int32_t (*buffer_len)(void* buffer, uint64_t* len);                                  /* ExperienceBufferBase::len */
    int32_t (*publish_model)(void* agent, void* mailbox, uint64_t n_opts);               /* AsyncTrainer::sync */
    void* mailbox;
    int32_t (*exchange)(void* ctx, void* agent, uint64_t opt_steps);                     /* optional: cross-GPU averaging at a sync point */
    void* exchange_ctx;
    /* optional, with `exchange`: agreement across ranks BEFORE every collective.  Called with local_ok = 1 at every sync point
     * (a rank that gets *all_ok = 0 stops there with BDR_ERR_COMM instead of entering a collective a failed peer will never
     * join) and once with local_ok = 0 by a rank whose learner or actor failed, so that its peers stop at their next sync point
     * (bdr_comm_agree: an RCCL MIN all-reduce of the flag).  ctx = exchange_ctx. */
    int32_t (*agree)(void* ctx, int32_t local_ok, int32_t* all_ok);
} bdr_learner_ops;

typedef struct bdr_actor_ops {              /* one Actor: its own agent (built from its own config) and environment */
    void* agent;
    void* mailbox;
    int32_t (*agent_set_train)(void* agent, int32_t train);
    int32_t (*agent_sample)(void* agent, uint64_t n_procs, const void* obs, void* act_out);
    int32_t (*sync_model)(void* agent, void* mailbox, uint32_t actor_id, int32_t first, uint64_t* n_opts_inout, int32_t* updated);
    int32_t (*agent_sample_device)(void* agent, uint64_t n_procs, const void* obs_dev, uint64_t row_stride, void* act_out);   /* env.obs_on_device */
    bdr_env_vtable env;
} bdr_actor_ops;

typedef struct bdr_async_stats {            /* AsyncTrainStat (async_trainer/stat.rs) + counters */
    uint64_t samples_total, opt_steps, n_records, n_syncs, n_messages;
    double duration_s;
    float samples_per_sec, opt_per_sec;
} bdr_async_stats;
typedef struct bdr_actor_stat { uint64_t env_steps, n_syncs; double duration_s; } bdr_actor_stat;   /* ActorStat (actor/stat.rs) */

/* observer(ctx, actor, a, b, event, scalars, n): never called concurrently.  actor = UINT32_MAX for learner events.
 *   SKIP / OPT / OPT_RECORD   a = samples_total, b = opt_steps after the step, scalars = the Record of opt_with_record
 *   SYNC                      the learner published the model of b = opt_steps
 *   PUSH                      a message of n transitions of `actor` entered the buffer (a = samples_total after it)
 *   COST                      scalars = {average_opt_time, average_sample_time} in ms
 *   ACTOR_SYNC                `actor` adopted the model of b opt steps before its env step a */
#define BDR_ASYNC_EVENT_SKIP 0
#define BDR_ASYNC_EVENT_OPT 1
#define BDR_ASYNC_EVENT_OPT_RECORD 2
#define BDR_ASYNC_EVENT_COST 3
#define BDR_ASYNC_EVENT_SYNC 4
#define BDR_ASYNC_EVENT_PUSH 5
#define BDR_ASYNC_EVENT_ACTOR_SYNC 6
typedef void (*bdr_async_observer)(void* ctx, uint32_t actor, uint64_t a, uint64_t b, int32_t event, const float* scalars, int32_t n);

BDR_API void bdr_async_trainer_config_default(bdr_async_trainer_config* c);
BDR_API void bdr_learner_ops_default(bdr_learner_ops* ops, bdr_agent* agent, bdr_replay* buffer, bdr_model_mailbox* mailbox);
BDR_API void bdr_actor_ops_default(bdr_actor_ops* ops, bdr_agent* agent, bdr_model_mailbox* mailbox, const bdr_env_vtable* env);
/* train_async (util.rs:31-92): runs until the learner has done max_opts opt steps; returns the first error of any thread. */
BDR_API int32_t bdr_async_train(const bdr_async_trainer_config* c, const bdr_learner_ops* learner, const bdr_actor_ops* actors,
                                uint32_t n_actors, bdr_async_observer observer, void* observer_ctx, bdr_async_stats* out,
                                bdr_actor_stat* actor_stats);

/* border-atari-env frame preprocessing on the device (border-atari-env/src/env.rs):
 * one handle keeps the `frames: [4][84][84]` u8 stack of n_envs environments in HBM (newest frame first).
 *   reset  env.rs:263-296   all four slots <- warp_and_grayscale(frame)
 *   step   env.rs:312-324   skip_and_max (:126-157: element-wise max of the two last RGB frames of the skip-4 step),
 *                           warp_and_grayscale (:171-195: image 0.23 resize(.., 84, 84, Triangle), then the reference's luma
 *                           with its (b, g, r) channel naming), stack_frame (:197-209)
 * frames are host buffers [n][height][width][3] u8 (render_rgb24 layout); env_ixs[k] names the environment of frame k and
 * may not repeat within one call.  The resize restates image 0.23.14's sample.rs (see oracle/atari_prep.py: parity
 * unpinned against the real crate).  bdr_atari_clip_reward: env.rs:159-169. */
typedef struct bdr_atari_prep bdr_atari_prep;
BDR_API int32_t bdr_atari_prep_create(int32_t device, uint32_t n_envs, uint32_t width, uint32_t height, bdr_atari_prep** out);
BDR_API int32_t bdr_atari_prep_destroy(bdr_atari_prep* h);
BDR_API int32_t bdr_atari_prep_reset(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, const uint8_t* frames);
BDR_API int32_t bdr_atari_prep_step(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, const uint8_t* frames_a,
                                    const uint8_t* frames_b);
/* stacked observations of the named environments, [n][4][84][84] u8, to the host (what BorderAtariObs carries) */
BDR_API int32_t bdr_atari_prep_obs(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, uint8_t* obs_out);
/* device address of all stacks, [n_envs][4][84][84] u8 */
BDR_API int32_t bdr_atari_prep_device_stacks(bdr_atari_prep* h, const uint8_t** stacks);
/* one environment's current stack -> dst_dev (device memory, 4*84*84 bytes), complete when the call returns: what an environment
 * with bdr_env_vtable::obs_on_device writes into obs_out / init_obs_out */
BDR_API int32_t bdr_atari_prep_copy_stack(bdr_atari_prep* h, uint32_t env_ix, void* dst_dev);
/* device address of every environment's stack as it was BEFORE its last step, [n_envs][4][84][84] u8: obs_t of the transition
 * whose next_obs is device_stacks (what bdr_replay_push_device takes; unchanged by reset) */
BDR_API int32_t bdr_atari_prep_device_prev_stacks(bdr_atari_prep* h, const uint8_t** prev);
BDR_API float bdr_atari_clip_reward(float r, int32_t train);

/* Host-side named-tensor files (no GPU involved): the container layer under save_params / load_params, for callers
 * that move parameters themselves (bdr_agent_get_params / set_params take the same reference-layout vectors).
 * `data` is the concatenation of the tensors in `meta` order (row-major f32); the container is chosen by the file
 * name as above.  Reading matches by name, checks shapes, ignores extra entries; nested module archives exported
 * from Python (torch.jit.save of a module with submodules c1, c2, ...) are flattened to dotted names. */
typedef struct bdr_named_tensor {
    const char* name;
    const uint64_t* dims;
    uint32_t ndim;
} bdr_named_tensor;
BDR_API int32_t bdr_checkpoint_write(const char* path, const bdr_named_tensor* meta, uint32_t n_tensors, const float* data, uint64_t n);
BDR_API int32_t bdr_checkpoint_read(const char* path, const bdr_named_tensor* meta, uint32_t n_tensors, float* data, uint64_t n);

/* Parity probes: copy intermediates of the LAST update to the host.
 * what: 0 q_pred_all [B][A], 1 q_next_all [B][A], 2 pred [B], 3 tgt [B], 4 loss [1];
 * AtariCnn only, the online network's activations on `obs`, position-major (NHWC): 5 conv1 [B][400][32], 6 conv2 [B][81][64], 7 conv3 [B][49][64]. */
BDR_API int32_t bdr_dqn_probe(bdr_agent* a, int32_t what, float* out, uint64_t n);

/* Per-kernel device timing of the opt step (bench.py roofline leg): when enabled the step is
 * bracketed kernel by kernel with HIP events on the agent's stream. */
BDR_API int32_t bdr_agent_profile_enable(bdr_agent* a, int32_t on);
/* names_out: '\n'-separated kernel labels; ms_out[i] = mean ms per launch since enable. */
BDR_API int32_t bdr_agent_profile_read(bdr_agent* a, char* names_out, uint64_t names_cap,
                                       float* ms_out, uint64_t* count_inout);

/* ------------------------------------------------------------------------------------------
 * IQN agent  (border-tch-agent/src/iqn/base.rs, iqn/config.rs:50-67, iqn/model/base.rs)
 * ---------------------------------------------------------------------------------------- */
/* IqnSample (iqn/model/base.rs:327-387); Const32 yields 33 points like the reference. */
enum { BDR_IQN_CONST10 = 0, BDR_IQN_CONST32 = 1, BDR_IQN_UNIFORM10 = 2, BDR_IQN_UNIFORM8 = 3,
       BDR_IQN_UNIFORM32 = 4, BDR_IQN_UNIFORM64 = 5, BDR_IQN_MEDIAN = 6 };
typedef struct {
    bdr_net_config psi;          /* feature extractor F: AtariCnn{skip_linear:true} or Mlp (out_dim = feature_dim) */
    int32_t feature_dim, embed_dim;                   /* IqnModelConfig */
    int32_t n_f_units; int32_t f_units[BDR_MAX_UNITS];   /* merge net M = Mlp(feature_dim -> units -> n_actions) */
    int32_t n_actions;
    double lr;                                        /* OptimizerConfig::{Adam,AdamW}.lr; AdamW fields: `opt` below */
    uint64_t soft_update_interval, n_updates_per_opt, batch_size;
    double discount_factor, tau;
    int32_t sample_percents_pred, sample_percents_tgt, sample_percents_act;
    int32_t train;
    int32_t device;
    uint64_t seed;
    bdr_adamw_config opt;        /* IqnModelConfig.opt_config (iqn/model/config.rs:50) when it is AdamW; `lr` above either way */
    int32_t arithmetic;          /* BDR_ARITH_* */
    int32_t reserved;
} bdr_iqn_config;
BDR_API void bdr_iqn_config_default(bdr_iqn_config* cfg);                    /* iqn/config.rs:50-67 */
BDR_API int32_t bdr_iqn_create(const bdr_iqn_config* cfg, bdr_agent** out);  /* iqn/base.rs:230-268 */
/* One Iqn::opt_ update on a host minibatch with injected percent points (the reference draws them with
 * Tensor::rand, iqn/model/base.rs:365-368). */
BDR_API int32_t bdr_iqn_update_on_batch(bdr_agent* a, uint64_t n, const void* obs, const int64_t* act,
                                        const void* next_obs, const float* reward, const int8_t* is_terminated,
                                        const float* tau_pred, int32_t n_pred, const float* tau_tgt, int32_t n_tgt,
                                        float* loss_out);
/* IqnModel::forward (iqn/model/base.rs:198-234): z_out [n][n_tau][n_actions]; which 0 = iqn, 1 = iqn_tgt. */
BDR_API int32_t bdr_iqn_forward(bdr_agent* a, int32_t which, uint64_t n, const void* obs, const float* tau,
                                int32_t n_tau, float* z_out);
/* Policy::sample greedy part (iqn/base.rs:204-228): values averaged over sample_percents_act. */
BDR_API int32_t bdr_iqn_qvalues(bdr_agent* a, uint64_t n, const void* obs, float* q_out, int64_t* argmax_out);

/* ------------------------------------------------------------------------------------------
 * SAC agent  (border-tch-agent/src/sac/base.rs, sac/config.rs:85-105, sac/ent_coef.rs,
 * Actor = Mlp2 (mlp/mlp2.rs), Critic = Mlp on cat(obs, act) (mlp/base.rs:83-107))
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t obs_dim, act_dim;
    int32_t n_pi_units; int32_t pi_units[BDR_MAX_UNITS];   /* ActorConfig.pi_config (MlpConfig.units) */
    int32_t n_q_units;  int32_t q_units[BDR_MAX_UNITS];    /* CriticConfig.q_config                  */
    double lr_actor, lr_critic;                             /* lr of each model; AdamW fields: opt_actor / opt_critic below */
    double gamma, tau;
    int32_t ent_coef_auto;       /* EntCoefMode::Auto(target_entropy, lr) vs Fix(alpha) */
    double ent_coef_alpha, target_entropy, ent_coef_lr;
    double epsilon, min_lstd, max_lstd;
    uint64_t n_updates_per_opt, batch_size;
    int32_t train;
    int32_t critic_loss;         /* BDR_LOSS_* */
    double reward_scale;
    int32_t n_critics;
    int32_t device;
    uint64_t seed;
    bdr_adamw_config opt_actor;  /* ActorConfig.opt_config (sac/actor/config.rs:15) when it is AdamW; lr_actor above either way  */
    bdr_adamw_config opt_critic; /* CriticConfig.opt_config (sac/critic/config.rs) likewise; EntCoef keeps nn::Adam::default() (ent_coef.rs:41) */
} bdr_sac_config;
BDR_API void bdr_sac_config_default(bdr_sac_config* cfg);                    /* sac/config.rs:85-105  */
BDR_API int32_t bdr_sac_create(const bdr_sac_config* cfg, bdr_agent** out);  /* sac/base.rs:237-285   */
/* One Sac::opt_ loop iteration on a host minibatch with injected N(0,1) draws (the reference takes
 * them from torch's global CPU generator, sac/base.rs:76).  rec3: loss_critic, loss_actor, ent_coef.
 * Parameter models for bdr_agent_{get,set}_params / param_count_of: 0 pi, 1+i qnet_i,
 * 1+n_critics+i qnet_tgt_i, 1+2*n_critics log_alpha; +100 gradient, +200 exp_avg, +300 exp_avg_sq, +400 max_exp_avg_sq (amsgrad). */
BDR_API int32_t bdr_sac_update_on_batch(bdr_agent* a, uint64_t n, const float* obs, const float* act,
                                        const float* next_obs, const float* reward, const int8_t* is_terminated,
                                        const float* z_actor, const float* z_next, float* rec3);
/* Parity probes: intermediates of the LAST SAC update, to the host.  what:
 *   0 q_pred [n_critics][B]   Q_i(obs, act), the predictions of update_critic (sac/base.rs:128-131; qvals :89-98)
 *   1 q_next [n_critics][B]   target critics on (next_obs, a'), a' ~ pi(next_obs) of the UPDATED actor (:112-118)
 *   2 qvals_min [B]           min over 1 (:100-105)          3 next_log_p [B]  log p(a' | next_obs) (:113)
 *   4 tgt [B]                 the TD target (:119-122)       7 next_act [B][act_dim]
 *   5 q_pi [n_critics][B]     Q_i(obs, a_pi) of update_actor (:157-158)      6 log_p [B]  log p(a_pi | obs) (:156)
 * 5 and 6 are overwritten by the critic phase and therefore kept aside by bdr_sac_update_on_batch only. */
BDR_API int32_t bdr_sac_probe(bdr_agent* a, int32_t what, float* out, uint64_t n);
/* Policy::sample (sac/base.rs:215-225). */
BDR_API int32_t bdr_sac_sample(bdr_agent* a, uint64_t n, const float* obs, float* act_out);
/* the same for observation rows in HBM (row i at obs_dev + i * row_stride bytes), see bdr_agent_sample_device */
BDR_API int32_t bdr_sac_sample_device(bdr_agent* a, uint64_t n, const void* obs_dev, uint64_t row_stride, float* act_out);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU parameter exchange (replaces the learner->actors NamedTensors channel of
 * border-async-trainer/src/async_trainer/base.rs:268-272 with RCCL over xGMI).
 * ---------------------------------------------------------------------------------------- */
#define BDR_UNIQUE_ID_BYTES 128
BDR_API int32_t bdr_comm_get_unique_id(uint8_t id[BDR_UNIQUE_ID_BYTES]);
BDR_API int32_t bdr_comm_init_rank(const uint8_t id[BDR_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank,
                                   int32_t device, bdr_comm** out);
BDR_API int32_t bdr_comm_destroy(bdr_comm* c);
/* *all_ok <- MIN over ranks of (local_ok != 0): the agreement every rank reaches before it enters a collective, so that a rank
 * whose learner failed does not leave its peers blocked in the next all-reduce (bdr_learner_ops::agree).  Synchronous. */
BDR_API int32_t bdr_comm_agree(bdr_comm* c, int32_t local_ok, int32_t* all_ok);
/* params <- mean over ranks (ncclAllReduce sum on the flat arena, then 1/nranks), on the
 * agent's stream; which as in bdr_agent_get_params (0 qnet, 1 qnet_tgt, 2/3 Adam moments). */
BDR_API int32_t bdr_agent_allreduce_params(bdr_agent* a, bdr_comm* c, int32_t which);
/* Synchronous data-parallel mode for DQN agents: from now on every Agent::opt of `a` runs backward, all-reduces the gradient
 * arena over `c` (ncclAllReduce sum, then 1/nranks, on the agent's stream) and then takes the optimizer step, so the ranks
 * stay bit-for-bit in lock step and N x batch B/N equals one step on batch B.  c == NULL: back to independent steps. */
BDR_API int32_t bdr_agent_set_grad_comm(bdr_agent* a, bdr_comm* c);
/* params <- root's (ncclBroadcast): the faithful learner->actor sync. */
BDR_API int32_t bdr_agent_broadcast_params(bdr_agent* a, bdr_comm* c, int32_t which, int32_t root);

#ifdef __cplusplus
}
#endif
#endif /* BORDER_AMD_H */
