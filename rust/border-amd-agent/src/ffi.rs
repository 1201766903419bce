//! Declarations of `include/border_amd.h`, one to one: every `#[repr(C)]` struct below has the header's fields in the header's
//! order with the header's types, every `extern "C"` function the header's arguments.  `tests/test_rust_shim_layout.py` of the
//! border_amd repository parses this file and the header and fails on any field / order / type / arity mismatch, so the two
//! cannot drift apart unnoticed (this image has no cargo: the check is textual, not a compile).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const BDR_OK: i32 = 0;
pub const BDR_ERR_INVALID: i32 = 1;
pub const BDR_ERR_NO_DEVICE: i32 = 2;
pub const BDR_ERR_HIP: i32 = 3;
pub const BDR_ERR_EMPTY: i32 = 4;
pub const BDR_ERR_IO: i32 = 5;
pub const BDR_ERR_COMM: i32 = 6;

pub const BDR_RNG_STDRNG: i32 = 0;
pub const BDR_RNG_XOSHIRO256PP: i32 = 1;
pub const BDR_PER_NORMALIZE_ALL: i32 = 0;
pub const BDR_PER_NORMALIZE_BATCH: i32 = 1;
pub const BDR_NET_ATARI_CNN: i32 = 0;
pub const BDR_NET_MLP: i32 = 1;
pub const BDR_LOSS_MSE: i32 = 0;
pub const BDR_LOSS_SMOOTH_L1: i32 = 1;
pub const BDR_OPT_ADAM: i32 = 0;
pub const BDR_OPT_ADAMW: i32 = 1;
pub const BDR_ARITH_BF16X3_6: i32 = 0;
pub const BDR_ARITH_F32_EXACT: i32 = 1;
pub const BDR_MAX_UNITS: usize = 8;
pub const BDR_EXPLORER_SOFTMAX: i32 = 0;
pub const BDR_EXPLORER_EPS_GREEDY: i32 = 1;
pub const BDR_CKPT_TCH: i32 = 0;
pub const BDR_CKPT_SAFETENSORS: i32 = 1;
pub const BDR_IQN_CONST10: i32 = 0;
pub const BDR_IQN_CONST32: i32 = 1;
pub const BDR_IQN_UNIFORM10: i32 = 2;
pub const BDR_IQN_UNIFORM8: i32 = 3;
pub const BDR_IQN_UNIFORM32: i32 = 4;
pub const BDR_IQN_UNIFORM64: i32 = 5;
pub const BDR_IQN_MEDIAN: i32 = 6;
pub const BDR_UNIQUE_ID_BYTES: usize = 128;
pub const BDR_ASYNC_EVENT_SKIP: i32 = 0;
pub const BDR_ASYNC_EVENT_OPT: i32 = 1;
pub const BDR_ASYNC_EVENT_OPT_RECORD: i32 = 2;
pub const BDR_ASYNC_EVENT_COST: i32 = 3;
pub const BDR_ASYNC_EVENT_SYNC: i32 = 4;
pub const BDR_ASYNC_EVENT_PUSH: i32 = 5;
pub const BDR_ASYNC_EVENT_ACTOR_SYNC: i32 = 6;
pub const BDR_TRAINER_EVENT_SKIP: i32 = 0;
pub const BDR_TRAINER_EVENT_OPT: i32 = 1;
pub const BDR_TRAINER_EVENT_OPT_RECORD: i32 = 2;
pub const BDR_TRAINER_EVENT_COST: i32 = 3;

// opaque handles
#[repr(C)]
pub struct bdr_replay {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bdr_agent {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bdr_comm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bdr_model_mailbox {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bdr_atari_prep {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_replay_config {
    pub capacity: u64,
    pub seed: u64,
    pub obs_row_bytes: u64,
    pub act_row_bytes: u64,
    pub device: i32,
    pub frame_stack: i32,
    pub frame_capacity: u64,
    pub index_rng: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_device_batch {
    pub n: u64,
    pub obs: *const c_void,
    pub next_obs: *const c_void,
    pub act: *const c_void,
    pub reward: *const f32,
    pub is_terminated: *const i8,
    pub is_truncated: *const i8,
    pub ixs: *const u64,
    pub weight: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_per_config {
    pub alpha: f32,
    pub beta_0: f32,
    pub beta_final: f32,
    pub n_opts_final: u64,
    pub normalize: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_per_info {
    pub n_samples: u64,
    pub n_opts: u64,
    pub beta: f32,
    pub total: f32,
    pub max_p: f32,
    pub min_p: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_net_config {
    pub kind: i32,
    pub n_stack: i32,
    pub in_dim: i32,
    pub n_units: i32,
    pub units: [i32; 8],
    pub out_dim: i32,
    pub activation_out: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_adamw_config {
    pub opt_kind: i32,
    pub amsgrad: i32,
    pub beta1: f64,
    pub beta2: f64,
    pub weight_decay: f64,
    pub eps: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_dqn_config {
    pub net: bdr_net_config,
    pub opt_kind: i32,
    pub lr: f64,
    pub beta1: f64,
    pub beta2: f64,
    pub weight_decay: f64,
    pub eps: f64,
    pub amsgrad: i32,
    pub soft_update_interval: u64,
    pub n_updates_per_opt: u64,
    pub batch_size: u64,
    pub discount_factor: f64,
    pub tau: f64,
    pub train: i32,
    pub double_dqn: i32,
    pub critic_loss: i32,
    pub has_clip_td_err: i32,
    pub clip_td_err_min: f64,
    pub clip_td_err_max: f64,
    pub record_verbose_level: i32,
    pub device: i32,
    pub param_seed: u64,
    pub arithmetic: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_dqn_record {
    pub loss: f32,
    pub pred_mean: f32,
    pub reward_mean: f32,
    pub tgt_mean: f32,
    pub tgt_minus_pred_mean: f32,
    pub has_verbose: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_explorer_config {
    pub kind: i32,
    pub eps_start: f64,
    pub eps_final: f64,
    pub final_step: u64,
    pub n_calls: u64,
    pub seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_sample_info {
    pub eps: f64,
    pub is_random: i32,
    pub n_samples_act: u64,
    pub n_samples_best_act: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bdr_env_vtable {
    pub ctx: *mut c_void,
    pub reset: Option<unsafe extern "C" fn(ctx: *mut c_void, obs_out: *mut c_void) -> i32>,
    pub step_with_reset: Option<
        unsafe extern "C" fn(
            ctx: *mut c_void,
            act: *const c_void,
            obs_out: *mut c_void,
            reward: *mut f32,
            is_terminated: *mut i8,
            is_truncated: *mut i8,
            init_obs_out: *mut c_void,
        ) -> i32,
    >,
    pub obs_on_device: i32,
    pub device: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bdr_trainer_ops {
    pub agent: *mut c_void,
    pub buffer: *mut c_void,
    pub agent_set_train: Option<unsafe extern "C" fn(agent: *mut c_void, train: i32) -> i32>,
    pub agent_sample: Option<unsafe extern "C" fn(agent: *mut c_void, n_procs: u64, obs: *const c_void, act_out: *mut c_void) -> i32>,
    pub agent_opt: Option<unsafe extern "C" fn(agent: *mut c_void, buffer: *mut c_void) -> i32>,
    pub agent_opt_with_record:
        Option<unsafe extern "C" fn(agent: *mut c_void, buffer: *mut c_void, scalars: *mut f32, cap: i32, n_scalars: *mut i32) -> i32>,
    pub buffer_push: Option<
        unsafe extern "C" fn(
            buffer: *mut c_void,
            n: u64,
            obs: *const c_void,
            act: *const c_void,
            next_obs: *const c_void,
            reward: *const f32,
            is_terminated: *const i8,
            is_truncated: *const i8,
        ) -> i32,
    >,
    pub agent_sample_device:
        Option<unsafe extern "C" fn(agent: *mut c_void, n_procs: u64, obs_dev: *const c_void, row_stride: u64, act_out: *mut c_void) -> i32>,
    pub buffer_push_device: Option<
        unsafe extern "C" fn(
            buffer: *mut c_void,
            n: u64,
            obs_dev: *const c_void,
            obs_stride: u64,
            act: *const c_void,
            next_obs_dev: *const c_void,
            next_obs_stride: u64,
            reward: *const f32,
            is_terminated: *const i8,
            is_truncated: *const i8,
        ) -> i32,
    >,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_trainer_config {
    pub max_opts: u64,
    pub opt_interval: u64,
    pub warmup_period: u64,
    pub record_agent_info_interval: u64,
    pub record_compute_cost_interval: u64,
    pub obs_row_bytes: u64,
    pub act_row_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_trainer_stats {
    pub env_steps: u64,
    pub opt_steps: u64,
    pub n_records: u64,
    pub n_episodes: u64,
    pub opt_seconds: f64,
    pub sample_seconds: f64,
}

pub type bdr_trainer_observer =
    Option<unsafe extern "C" fn(ctx: *mut c_void, env_steps: u64, opt_steps: u64, event: i32, scalars: *const f32, n_scalars: i32)>;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_async_trainer_config {
    pub max_opts: u64,
    pub warmup_period: u64,
    pub sync_interval: u64,
    pub record_agent_info_interval: u64,
    pub record_compute_cost_interval: u64,
    pub n_buffer: u64,
    pub channel_capacity: u64,
    pub warmup_sleep_ms: u64,
    pub obs_row_bytes: u64,
    pub act_row_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bdr_learner_ops {
    pub t: bdr_trainer_ops,
    pub buffer_len: Option<unsafe extern "C" fn(buffer: *mut c_void, len: *mut u64) -> i32>,
    pub publish_model: Option<unsafe extern "C" fn(agent: *mut c_void, mailbox: *mut c_void, n_opts: u64) -> i32>,
    pub mailbox: *mut c_void,
    pub exchange: Option<unsafe extern "C" fn(ctx: *mut c_void, agent: *mut c_void, opt_steps: u64) -> i32>,
    pub exchange_ctx: *mut c_void,
    pub agree: Option<unsafe extern "C" fn(ctx: *mut c_void, local_ok: i32, all_ok: *mut i32) -> i32>,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bdr_actor_ops {
    pub agent: *mut c_void,
    pub mailbox: *mut c_void,
    pub agent_set_train: Option<unsafe extern "C" fn(agent: *mut c_void, train: i32) -> i32>,
    pub agent_sample: Option<unsafe extern "C" fn(agent: *mut c_void, n_procs: u64, obs: *const c_void, act_out: *mut c_void) -> i32>,
    pub sync_model: Option<
        unsafe extern "C" fn(agent: *mut c_void, mailbox: *mut c_void, actor_id: u32, first: i32, n_opts_inout: *mut u64, updated: *mut i32) -> i32,
    >,
    pub agent_sample_device:
        Option<unsafe extern "C" fn(agent: *mut c_void, n_procs: u64, obs_dev: *const c_void, row_stride: u64, act_out: *mut c_void) -> i32>,
    pub env: bdr_env_vtable,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_async_stats {
    pub samples_total: u64,
    pub opt_steps: u64,
    pub n_records: u64,
    pub n_syncs: u64,
    pub n_messages: u64,
    pub duration_s: f64,
    pub samples_per_sec: f32,
    pub opt_per_sec: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bdr_actor_stat {
    pub env_steps: u64,
    pub n_syncs: u64,
    pub duration_s: f64,
}

pub type bdr_async_observer =
    Option<unsafe extern "C" fn(ctx: *mut c_void, actor: u32, a: u64, b: u64, event: i32, scalars: *const f32, n: i32)>;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct bdr_named_tensor {
    pub name: *const c_char,
    pub dims: *const u64,
    pub ndim: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_iqn_config {
    pub psi: bdr_net_config,
    pub feature_dim: i32,
    pub embed_dim: i32,
    pub n_f_units: i32,
    pub f_units: [i32; 8],
    pub n_actions: i32,
    pub lr: f64,
    pub soft_update_interval: u64,
    pub n_updates_per_opt: u64,
    pub batch_size: u64,
    pub discount_factor: f64,
    pub tau: f64,
    pub sample_percents_pred: i32,
    pub sample_percents_tgt: i32,
    pub sample_percents_act: i32,
    pub train: i32,
    pub device: i32,
    pub seed: u64,
    pub opt: bdr_adamw_config,
    pub arithmetic: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bdr_sac_config {
    pub obs_dim: i32,
    pub act_dim: i32,
    pub n_pi_units: i32,
    pub pi_units: [i32; 8],
    pub n_q_units: i32,
    pub q_units: [i32; 8],
    pub lr_actor: f64,
    pub lr_critic: f64,
    pub gamma: f64,
    pub tau: f64,
    pub ent_coef_auto: i32,
    pub ent_coef_alpha: f64,
    pub target_entropy: f64,
    pub ent_coef_lr: f64,
    pub epsilon: f64,
    pub min_lstd: f64,
    pub max_lstd: f64,
    pub n_updates_per_opt: u64,
    pub batch_size: u64,
    pub train: i32,
    pub critic_loss: i32,
    pub reward_scale: f64,
    pub n_critics: i32,
    pub device: i32,
    pub seed: u64,
    pub opt_actor: bdr_adamw_config,
    pub opt_critic: bdr_adamw_config,
}

#[link(name = "border_amd")]
extern "C" {
    pub fn bdr_last_error() -> *const c_char;
    pub fn bdr_last_error_is_deferred() -> i32;
    pub fn bdr_device_count(count: *mut i32) -> i32;
    pub fn bdr_version() -> *const c_char;

    // ---- SimpleReplayBuffer
    pub fn bdr_replay_create(cfg: *const bdr_replay_config, out: *mut *mut bdr_replay) -> i32;
    pub fn bdr_replay_destroy(r: *mut bdr_replay) -> i32;
    pub fn bdr_replay_push(
        r: *mut bdr_replay,
        n: u64,
        obs: *const c_void,
        act: *const c_void,
        next_obs: *const c_void,
        reward: *const f32,
        is_terminated: *const i8,
        is_truncated: *const i8,
    ) -> i32;
    pub fn bdr_replay_push_device(
        r: *mut bdr_replay,
        n: u64,
        obs_dev: *const c_void,
        obs_stride: u64,
        act: *const c_void,
        next_obs_dev: *const c_void,
        next_obs_stride: u64,
        reward: *const f32,
        is_terminated: *const i8,
        is_truncated: *const i8,
    ) -> i32;
    pub fn bdr_replay_len(r: *const bdr_replay, len: *mut u64) -> i32;
    pub fn bdr_replay_head(r: *const bdr_replay, head: *mut u64) -> i32;
    pub fn bdr_replay_frames_used(r: *const bdr_replay, allocated: *mut u64, capacity: *mut u64) -> i32;
    pub fn bdr_replay_sample_indices(r: *mut bdr_replay, n: u64, ixs_out: *mut u64) -> i32;
    pub fn bdr_replay_batch(
        r: *mut bdr_replay,
        n: u64,
        ixs_out: *mut u64,
        obs_out: *mut c_void,
        act_out: *mut c_void,
        next_obs_out: *mut c_void,
        reward_out: *mut f32,
        is_terminated_out: *mut i8,
        is_truncated_out: *mut i8,
    ) -> i32;
    pub fn bdr_replay_last_batch(r: *const bdr_replay, out: *mut bdr_device_batch) -> i32;
    pub fn bdr_replay_fill_synthetic(r: *mut bdr_replay, n: u64, seed: u64, kind: i32, n_actions: i32) -> i32;
    pub fn bdr_replay_read_rows(
        r: *mut bdr_replay,
        first: u64,
        n: u64,
        obs: *mut c_void,
        act: *mut c_void,
        next_obs: *mut c_void,
        reward: *mut f32,
        term: *mut i8,
        trunc: *mut i8,
    ) -> i32;

    // ---- prioritized replay
    pub fn bdr_per_config_default(c: *mut bdr_per_config);
    pub fn bdr_replay_enable_per(r: *mut bdr_replay, c: *const bdr_per_config) -> i32;
    pub fn bdr_replay_update_priority(r: *mut bdr_replay, n: u64, ixs: *const u64, td_errs: *const f32) -> i32;
    pub fn bdr_replay_batch_weights(r: *mut bdr_replay, n: u64, w_out: *mut f32) -> i32;
    pub fn bdr_replay_per_info(r: *mut bdr_replay, out: *mut bdr_per_info) -> i32;
    pub fn bdr_replay_per_read(r: *mut bdr_replay, what: i32, out: *mut f32, n: u64) -> i32;
    pub fn bdr_replay_per_get(r: *mut bdr_replay, s: f32, ix: *mut u64) -> i32;

    // ---- DQN and the agent surface shared by every kind
    pub fn bdr_dqn_config_default(cfg: *mut bdr_dqn_config);
    pub fn bdr_dqn_create(cfg: *const bdr_dqn_config, out: *mut *mut bdr_agent) -> i32;
    pub fn bdr_agent_destroy(a: *mut bdr_agent) -> i32;
    pub fn bdr_agent_set_train(a: *mut bdr_agent, train: i32) -> i32;
    pub fn bdr_agent_is_train(a: *const bdr_agent, out: *mut i32) -> i32;
    pub fn bdr_agent_opt(a: *mut bdr_agent, buffer: *mut bdr_replay) -> i32;
    pub fn bdr_agent_opt_with_record(a: *mut bdr_agent, buffer: *mut bdr_replay, rec: *mut bdr_dqn_record) -> i32;
    pub fn bdr_agent_opt_with_scalars(a: *mut bdr_agent, buffer: *mut bdr_replay, out: *mut f32, cap: i32, n_out: *mut i32) -> i32;
    pub fn bdr_agent_record_keys(a: *mut bdr_agent, names_out: *mut c_char, names_cap: u64, n_keys: *mut i32) -> i32;
    pub fn bdr_agent_draw_noise(a: *mut bdr_agent, n: u64, out: *mut f32) -> i32;
    pub fn bdr_dqn_update_on_batch(
        a: *mut bdr_agent,
        n: u64,
        obs: *const c_void,
        act: *const i64,
        next_obs: *const c_void,
        reward: *const f32,
        is_terminated: *const i8,
        rec: *mut bdr_dqn_record,
    ) -> i32;
    pub fn bdr_dqn_update_on_batch_weighted(
        a: *mut bdr_agent,
        n: u64,
        obs: *const c_void,
        act: *const i64,
        next_obs: *const c_void,
        reward: *const f32,
        is_terminated: *const i8,
        weight: *const f32,
        td_errs_out: *mut f32,
        rec: *mut bdr_dqn_record,
    ) -> i32;
    pub fn bdr_dqn_grads_on_batch(
        a: *mut bdr_agent,
        n: u64,
        obs: *const c_void,
        act: *const i64,
        next_obs: *const c_void,
        reward: *const f32,
        is_terminated: *const i8,
        rec: *mut bdr_dqn_record,
    ) -> i32;
    pub fn bdr_agent_apply_grads(a: *mut bdr_agent) -> i32;
    pub fn bdr_agent_qvalues(a: *mut bdr_agent, n: u64, obs: *const c_void, q_out: *mut f32, argmax_out: *mut i64) -> i32;
    pub fn bdr_explorer_config_default(e: *mut bdr_explorer_config, kind: i32);
    pub fn bdr_agent_set_explorer(a: *mut bdr_agent, e: *const bdr_explorer_config) -> i32;
    pub fn bdr_agent_get_explorer(a: *const bdr_agent, e: *mut bdr_explorer_config) -> i32;
    pub fn bdr_agent_sample(a: *mut bdr_agent, n_procs: u64, obs: *const c_void, act_out: *mut i64, info: *mut bdr_sample_info) -> i32;
    pub fn bdr_agent_sample_device(a: *mut bdr_agent, n_procs: u64, obs_dev: *const c_void, row_stride: u64, act_out: *mut i64, info: *mut bdr_sample_info) -> i32;
    pub fn bdr_agent_qvalues_device(a: *mut bdr_agent, n: u64, obs_dev: *const c_void, row_stride: u64, q_out: *mut f32, argmax_out: *mut i64) -> i32;
    pub fn bdr_agent_sync(a: *mut bdr_agent) -> i32;
    pub fn bdr_agent_n_opts(a: *const bdr_agent, n: *mut u64) -> i32;
    pub fn bdr_agent_param_count(a: *const bdr_agent, n: *mut u64) -> i32;
    pub fn bdr_agent_param_count_of(a: *mut bdr_agent, which: i32, n: *mut u64) -> i32;
    pub fn bdr_agent_get_params(a: *mut bdr_agent, which: i32, out: *mut f32, n: u64) -> i32;
    pub fn bdr_agent_set_params(a: *mut bdr_agent, which: i32, inp: *const f32, n: u64) -> i32;
    pub fn bdr_agent_arena_device_ptr(a: *mut bdr_agent, which: i32, ptr: *mut *mut c_void, n_floats: *mut u64) -> i32;
    pub fn bdr_agent_arena_release(a: *mut bdr_agent, which: i32) -> i32;
    pub fn bdr_agent_set_checkpoint_format(a: *mut bdr_agent, format: i32) -> i32;
    pub fn bdr_agent_save_params(a: *mut bdr_agent, dir: *const c_char) -> i32;
    pub fn bdr_agent_load_params(a: *mut bdr_agent, dir: *const c_char) -> i32;

    // ---- compiled Trainer loops
    pub fn bdr_trainer_config_default(c: *mut bdr_trainer_config);
    pub fn bdr_trainer_ops_default(ops: *mut bdr_trainer_ops, agent: *mut bdr_agent, buffer: *mut bdr_replay);
    pub fn bdr_trainer_train(
        c: *const bdr_trainer_config,
        ops: *const bdr_trainer_ops,
        env: *const bdr_env_vtable,
        observer: bdr_trainer_observer,
        observer_ctx: *mut c_void,
        out: *mut bdr_trainer_stats,
    ) -> i32;
    pub fn bdr_trainer_train_offline(
        c: *const bdr_trainer_config,
        ops: *const bdr_trainer_ops,
        observer: bdr_trainer_observer,
        observer_ctx: *mut c_void,
        out: *mut bdr_trainer_stats,
    ) -> i32;

    // ---- async trainer
    pub fn bdr_model_mailbox_create(device: i32, n_floats: u64, n_readers: u32, out: *mut *mut bdr_model_mailbox) -> i32;
    pub fn bdr_model_mailbox_destroy(m: *mut bdr_model_mailbox) -> i32;
    pub fn bdr_agent_publish_model(a: *mut bdr_agent, which: i32, m: *mut bdr_model_mailbox, n_opts: u64) -> i32;
    pub fn bdr_agent_sync_model_from(
        a: *mut bdr_agent,
        which: i32,
        m: *mut bdr_model_mailbox,
        reader: u32,
        first: i32,
        n_opts_inout: *mut u64,
        updated: *mut i32,
    ) -> i32;
    pub fn bdr_async_trainer_config_default(c: *mut bdr_async_trainer_config);
    pub fn bdr_learner_ops_default(ops: *mut bdr_learner_ops, agent: *mut bdr_agent, buffer: *mut bdr_replay, mailbox: *mut bdr_model_mailbox);
    pub fn bdr_actor_ops_default(ops: *mut bdr_actor_ops, agent: *mut bdr_agent, mailbox: *mut bdr_model_mailbox, env: *const bdr_env_vtable);
    pub fn bdr_async_train(
        c: *const bdr_async_trainer_config,
        learner: *const bdr_learner_ops,
        actors: *const bdr_actor_ops,
        n_actors: u32,
        observer: bdr_async_observer,
        observer_ctx: *mut c_void,
        out: *mut bdr_async_stats,
        actor_stats: *mut bdr_actor_stat,
    ) -> i32;

    // ---- Atari frame preprocessing
    pub fn bdr_atari_prep_create(device: i32, n_envs: u32, width: u32, height: u32, out: *mut *mut bdr_atari_prep) -> i32;
    pub fn bdr_atari_prep_destroy(h: *mut bdr_atari_prep) -> i32;
    pub fn bdr_atari_prep_reset(h: *mut bdr_atari_prep, n: u32, env_ixs: *const u32, frames: *const u8) -> i32;
    pub fn bdr_atari_prep_step(h: *mut bdr_atari_prep, n: u32, env_ixs: *const u32, frames_a: *const u8, frames_b: *const u8) -> i32;
    pub fn bdr_atari_prep_obs(h: *mut bdr_atari_prep, n: u32, env_ixs: *const u32, obs_out: *mut u8) -> i32;
    pub fn bdr_atari_prep_device_stacks(h: *mut bdr_atari_prep, stacks: *mut *const u8) -> i32;
    pub fn bdr_atari_prep_copy_stack(h: *mut bdr_atari_prep, env_ix: u32, dst_dev: *mut c_void) -> i32;
    pub fn bdr_atari_prep_device_prev_stacks(h: *mut bdr_atari_prep, prev: *mut *const u8) -> i32;
    pub fn bdr_atari_clip_reward(r: f32, train: i32) -> f32;

    // ---- checkpoints, probes, profiling
    pub fn bdr_checkpoint_write(path: *const c_char, meta: *const bdr_named_tensor, n_tensors: u32, data: *const f32, n: u64) -> i32;
    pub fn bdr_checkpoint_read(path: *const c_char, meta: *const bdr_named_tensor, n_tensors: u32, data: *mut f32, n: u64) -> i32;
    pub fn bdr_dqn_probe(a: *mut bdr_agent, what: i32, out: *mut f32, n: u64) -> i32;
    pub fn bdr_sac_probe(a: *mut bdr_agent, what: i32, out: *mut f32, n: u64) -> i32;
    pub fn bdr_agent_profile_enable(a: *mut bdr_agent, on: i32) -> i32;
    pub fn bdr_agent_profile_read(a: *mut bdr_agent, names_out: *mut c_char, names_cap: u64, ms_out: *mut f32, count_inout: *mut u64) -> i32;

    // ---- IQN
    pub fn bdr_iqn_config_default(cfg: *mut bdr_iqn_config);
    pub fn bdr_iqn_create(cfg: *const bdr_iqn_config, out: *mut *mut bdr_agent) -> i32;
    pub fn bdr_iqn_update_on_batch(
        a: *mut bdr_agent,
        n: u64,
        obs: *const c_void,
        act: *const i64,
        next_obs: *const c_void,
        reward: *const f32,
        is_terminated: *const i8,
        tau_pred: *const f32,
        n_pred: i32,
        tau_tgt: *const f32,
        n_tgt: i32,
        loss_out: *mut f32,
    ) -> i32;
    pub fn bdr_iqn_forward(a: *mut bdr_agent, which: i32, n: u64, obs: *const c_void, tau: *const f32, n_tau: i32, z_out: *mut f32) -> i32;
    pub fn bdr_iqn_qvalues(a: *mut bdr_agent, n: u64, obs: *const c_void, q_out: *mut f32, argmax_out: *mut i64) -> i32;

    // ---- SAC
    pub fn bdr_sac_config_default(cfg: *mut bdr_sac_config);
    pub fn bdr_sac_create(cfg: *const bdr_sac_config, out: *mut *mut bdr_agent) -> i32;
    pub fn bdr_sac_update_on_batch(
        a: *mut bdr_agent,
        n: u64,
        obs: *const f32,
        act: *const f32,
        next_obs: *const f32,
        reward: *const f32,
        is_terminated: *const i8,
        z_actor: *const f32,
        z_next: *const f32,
        rec3: *mut f32,
    ) -> i32;
    pub fn bdr_sac_sample(a: *mut bdr_agent, n: u64, obs: *const f32, act_out: *mut f32) -> i32;
    pub fn bdr_sac_sample_device(a: *mut bdr_agent, n: u64, obs_dev: *const c_void, row_stride: u64, act_out: *mut f32) -> i32;

    // ---- multi-GPU parameter exchange (RCCL over xGMI)
    pub fn bdr_comm_get_unique_id(id: *mut u8) -> i32;
    pub fn bdr_comm_init_rank(id: *const u8, nranks: i32, rank: i32, device: i32, out: *mut *mut bdr_comm) -> i32;
    pub fn bdr_comm_destroy(c: *mut bdr_comm) -> i32;
    pub fn bdr_comm_agree(c: *mut bdr_comm, local_ok: i32, all_ok: *mut i32) -> i32;
    pub fn bdr_agent_allreduce_params(a: *mut bdr_agent, c: *mut bdr_comm, which: i32) -> i32;
    pub fn bdr_agent_set_grad_comm(a: *mut bdr_agent, c: *mut bdr_comm) -> i32;
    pub fn bdr_agent_broadcast_params(a: *mut bdr_agent, c: *mut bdr_comm, which: i32, root: i32) -> i32;
}
