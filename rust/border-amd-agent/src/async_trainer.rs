//! `train_async` (`border-async-trainer/src/util.rs:31-92`) on the library's compiled loop (`bdr_async_train`).
//!
//! One rank = one GPU = one learner (`AsyncTrainer::train`, `async_trainer/base.rs:299-388`) + its actors (`Actor::run`,
//! `actor/base.rs:120-178`, one OS thread each, created by the library) + its local replay shard.  The learner -> actors channel
//! of the reference (`Arc<Mutex<(usize, ModelInfo)>>` fed by a crossbeam channel, `actor_manager/base.rs:112-129`) is a
//! device-resident mailbox; actors -> learner is the library's bounded(1000) queue of `n_buffer`-transition messages
//! (`ReplayBufferProxy::push`, `replay_buffer_proxy.rs:52-72`).  With a [`Comm`] the learners of all ranks are averaged over RCCL
//! at every sync point, after all ranks have agreed that every one of them is still running.
//!
//! The environments stay Rust: each actor's `E` is built in its own thread, as in the reference, and reached from the compiled
//! loop through two `extern "C"` trampolines around `Env::reset` / `Env::step_with_reset`.
use crate::{
    bytes::{ObsRows, RowBatch},
    comm::Comm,
    error::check,
    ffi,
    replay::AmdReplayBuffer,
    AmdDqn, AmdIqn, AmdSac,
};
use anyhow::Result;
use border_async_trainer::{ActorManagerConfig, ActorStat, AsyncTrainStat, AsyncTrainerConfig, SyncModel};
use border_core::{
    generic_replay_buffer::SimpleReplayBufferConfig,
    record::{Record, RecordValue},
    Agent, Configurable, Env, ReplayBufferBase,
};
use std::{os::raw::c_void, time::Duration};

/// What `bdr_async_train` needs from an agent of this crate.
pub trait AmdAgent<E: Env>: Configurable + SyncModel {
    fn raw(&self) -> *mut ffi::bdr_agent;
    /// Bytes of one action row as `Policy::sample` writes it (discrete: 8, one i64; SAC: `4 * act_dim`).
    fn act_row_bytes(&self) -> usize;
    /// The environment's action from one such row.
    fn decode_act(row: &[u8]) -> E::Act;
}

fn i64_of(row: &[u8]) -> i64 {
    let mut b = [0u8; 8];
    b.copy_from_slice(&row[..8]);
    i64::from_ne_bytes(b)
}

fn f32s_of(row: &[u8]) -> Vec<f32> {
    row.chunks_exact(4).map(|c| f32::from_ne_bytes([c[0], c[1], c[2], c[3]])).collect()
}

impl<E, O, A> AmdAgent<E> for AmdDqn<E, O, A>
where
    E: Env,
    E::Act: crate::bytes::DiscreteAct,
    O: RowBatch,
    A: RowBatch,
{
    fn raw(&self) -> *mut ffi::bdr_agent {
        self.handle()
    }
    fn act_row_bytes(&self) -> usize {
        8
    }
    fn decode_act(row: &[u8]) -> E::Act {
        <E::Act as crate::bytes::ActFromRows<i64>>::from_rows(vec![i64_of(row)], 1)
    }
}

impl<E, O, A> AmdAgent<E> for AmdIqn<E, O, A>
where
    E: Env,
    E::Act: crate::bytes::DiscreteAct,
    O: RowBatch,
    A: RowBatch,
{
    fn raw(&self) -> *mut ffi::bdr_agent {
        self.handle()
    }
    fn act_row_bytes(&self) -> usize {
        8
    }
    fn decode_act(row: &[u8]) -> E::Act {
        <E::Act as crate::bytes::ActFromRows<i64>>::from_rows(vec![i64_of(row)], 1)
    }
}

impl<E, O, A> AmdAgent<E> for AmdSac<E, O, A>
where
    E: Env,
    E::Act: crate::bytes::FloatAct,
    O: RowBatch,
    A: RowBatch,
{
    fn raw(&self) -> *mut ffi::bdr_agent {
        self.handle()
    }
    fn act_row_bytes(&self) -> usize {
        A::ROW_BYTES
    }
    fn decode_act(row: &[u8]) -> E::Act {
        <E::Act as crate::bytes::ActFromRows<f32>>::from_rows(f32s_of(row), 1)
    }
}

/// `AsyncTrainStat` + the counters the library adds.
#[derive(Debug, Clone)]
pub struct AmdAsyncTrainStat {
    pub stat: AsyncTrainStat,
    pub actor_stats: Vec<ActorStat>,
    pub samples_total: usize,
    pub opt_steps: usize,
    pub n_syncs: usize,
    pub n_messages: usize,
}

// ---- Env behind bdr_env_vtable ------------------------------------------------------------------------------------------------
struct EnvCtx<E: Env, A: AmdAgent<E>> {
    env: Option<E>,
    config: E::Config,
    seed: i64,
    obs_row_bytes: usize,
    act_row_bytes: usize,
    _a: std::marker::PhantomData<A>,
}

fn write_obs<Ob: ObsRows>(obs: &Ob, out: *mut c_void, row_bytes: usize) -> i32 {
    let b = obs.as_bytes();
    if obs.n_procs() != 1 || b.len() != row_bytes {
        return ffi::BDR_ERR_INVALID; // one process per actor, like Actor::run (step.is_terminated.len() == 1, env.rs:141)
    }
    // SAFETY: the compiled loop hands a buffer of obs_row_bytes bytes.
    unsafe { std::ptr::copy_nonoverlapping(b.as_ptr(), out as *mut u8, row_bytes) };
    ffi::BDR_OK
}

/// `Env::reset(None)`; the environment itself is built here, on the actor's thread, on the first call (`actor/base.rs:131-137`).
unsafe extern "C" fn env_reset<E, A>(ctx: *mut c_void, obs_out: *mut c_void) -> i32
where
    E: Env,
    E::Obs: ObsRows,
    A: AmdAgent<E>,
{
    let ctx = &mut *(ctx as *mut EnvCtx<E, A>);
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
        if ctx.env.is_none() {
            match E::build(&ctx.config, ctx.seed) {
                Ok(e) => ctx.env = Some(e),
                Err(_) => return 90,
            }
        }
        match ctx.env.as_mut().unwrap().reset(None) {
            Ok(obs) => write_obs(&obs, obs_out, ctx.obs_row_bytes),
            Err(_) => 90,
        }
    }));
    r.unwrap_or(90) // a panic must not unwind into C
}

/// `Env::step_with_reset(&act)` (`env.rs:137-161`).
unsafe extern "C" fn env_step<E, A>(
    ctx: *mut c_void,
    act: *const c_void,
    obs_out: *mut c_void,
    reward: *mut f32,
    is_terminated: *mut i8,
    is_truncated: *mut i8,
    init_obs_out: *mut c_void,
) -> i32
where
    E: Env,
    E::Obs: ObsRows,
    A: AmdAgent<E>,
{
    let ctx = &mut *(ctx as *mut EnvCtx<E, A>);
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
        let env = match ctx.env.as_mut() {
            Some(e) => e,
            None => return 91,
        };
        let row = std::slice::from_raw_parts(act as *const u8, ctx.act_row_bytes);
        let a = A::decode_act(row);
        let (step, _record) = env.step_with_reset(&a);
        let rc = write_obs(&step.obs, obs_out, ctx.obs_row_bytes);
        if rc != ffi::BDR_OK {
            return rc;
        }
        *reward = step.reward[0];
        *is_terminated = step.is_terminated[0];
        *is_truncated = step.is_truncated[0];
        if step.is_done() {
            match &step.init_obs {
                Some(o) => return write_obs(o, init_obs_out, ctx.obs_row_bytes),
                None => return 91,
            }
        }
        ffi::BDR_OK
    }));
    r.unwrap_or(91)
}

// ---- hooks ---------------------------------------------------------------------------------------------------------------------
struct ExchangeCtx<'a> {
    comm: &'a Comm,
    which: i32,
}

unsafe extern "C" fn exchange_cb(ctx: *mut c_void, agent: *mut c_void, _opt_steps: u64) -> i32 {
    let x = &*(ctx as *const ExchangeCtx);
    ffi::bdr_agent_allreduce_params(agent as *mut ffi::bdr_agent, x.comm.c, x.which)
}

unsafe extern "C" fn agree_cb(ctx: *mut c_void, local_ok: i32, all_ok: *mut i32) -> i32 {
    let x = &*(ctx as *const ExchangeCtx);
    ffi::bdr_comm_agree(x.comm.c, local_ok, all_ok)
}

struct ObserverCtx<'a> {
    keys: Vec<String>,
    recorder: &'a mut dyn FnMut(usize, Record),
}

/// What the reference hands its `Recorder` (`async_trainer/base.rs:224-266, 354-359`): the agent's `Record` every
/// `record_agent_info_interval` opts and the compute-cost averages every `record_compute_cost_interval`.
unsafe extern "C" fn observer_cb(ctx: *mut c_void, _actor: u32, _a: u64, opt_steps: u64, event: i32, scalars: *const f32, n: i32) {
    let o = &mut *(ctx as *mut ObserverCtx);
    let vals = if scalars.is_null() || n <= 0 { &[][..] } else { std::slice::from_raw_parts(scalars, n as usize) };
    let _ = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| match event {
        ffi::BDR_ASYNC_EVENT_OPT_RECORD => {
            let mut r = Record::empty();
            for (k, v) in o.keys.iter().zip(vals.iter()) {
                r.insert(k.clone(), RecordValue::Scalar(*v));
            }
            (o.recorder)(opt_steps as usize, r);
        }
        ffi::BDR_ASYNC_EVENT_COST if vals.len() == 2 => {
            let mut r = Record::empty();
            r.insert("average_opt_time", RecordValue::Scalar(vals[0]));
            r.insert("average_sample_time", RecordValue::Scalar(vals[1]));
            (o.recorder)(opt_steps as usize, r);
        }
        _ => {}
    }));
}

/// The device mailbox between a learner and its actors, destroyed on every exit path of `train_async` (an early `?` included).
/// Declared BEFORE the actors in `train_async`, so it is dropped after them: the actors' streams may still hold copies out of it
/// until their handles have synchronised and gone.
struct Mailbox(*mut ffi::bdr_model_mailbox);
impl Drop for Mailbox {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { ffi::bdr_model_mailbox_destroy(self.0) };
        }
    }
}

fn record_keys(agent: *mut ffi::bdr_agent) -> Result<Vec<String>> {
    let mut buf = vec![0u8; 16384];
    let mut n = 0i32;
    check(unsafe { ffi::bdr_agent_record_keys(agent, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len() as u64, &mut n) })?;
    let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
    Ok(String::from_utf8_lossy(&buf[..end]).split('\n').filter(|s| !s.is_empty()).map(|s| s.to_string()).collect())
}

/// `train_async` (`util.rs:31-92`): builds the learner's agent and replay buffer and one agent per entry of `agent_configs`
/// (`A::build(config.clone())`, `actor_manager/base.rs:141-175`; actor `i`'s environment is `E::build(env_config_train, i)`), runs
/// until the learner has done `max_opts` opt steps, and returns the statistics the reference logs.
///
/// * `recorder(opt_steps, record)` receives what the reference stores through its `Recorder`; evaluation and model saving
///   (`post_process`, `:224-266`) are the caller's (they need the reference's `Evaluator` / `Recorder`, untouched by this crate).
/// * `comm`: `Some(_)` when this process is one rank of several - the learners are averaged (`ncclAllReduce / n_ranks`) at every
///   sync point, right before the local publish.
#[allow(clippy::too_many_arguments)]
pub fn train_async<A, E, O, Ab>(
    agent_config: &A::Config,
    agent_configs: &Vec<A::Config>,
    env_config_train: &E::Config,
    replay_buffer_config: &SimpleReplayBufferConfig,
    actor_man_config: &ActorManagerConfig,
    async_trainer_config: &AsyncTrainerConfig,
    device: i32,
    recorder: &mut dyn FnMut(usize, Record),
    comm: Option<&Comm>,
) -> Result<(A, AmdAsyncTrainStat)>
where
    A: AmdAgent<E> + Agent<E, AmdReplayBuffer<O, Ab>> + 'static,
    A::Config: Clone,
    E: Env + 'static,
    E::Obs: ObsRows,
    O: RowBatch + 'static,
    Ab: RowBatch + 'static,
{
    // learner: agent + buffer (async_trainer/base.rs:160-188)
    let mut learner = A::build(agent_config.clone());
    let buffer = AmdReplayBuffer::<O, Ab>::build_on(
        replay_buffer_config,
        crate::replay::AmdReplayPlacement { device, frame_stack: 0, frame_capacity: 0, xoshiro_indices: false },
    )?;
    let _ = <AmdReplayBuffer<O, Ab> as ReplayBufferBase>::build; // (same Config type as the reference's R)
    let act_row_bytes = learner.act_row_bytes();

    let mut mailbox_guard = Mailbox(std::ptr::null_mut());   // (first: dropped last)
    // one agent per actor, each from its own config (actor/base.rs:127)
    let actors: Vec<A> = agent_configs.iter().map(|c| A::build(c.clone())).collect();
    let n_actors = actors.len();

    // the model channel: sized by the arena the agent's SyncModel ships (DQN qnet, IQN iqn, SAC pi = model 0)
    let mut n_floats = 0u64;
    let mut dev_ptr = std::ptr::null_mut();
    check(unsafe { ffi::bdr_agent_arena_device_ptr(learner.raw(), 0, &mut dev_ptr, &mut n_floats) })?;
    // only the size was wanted: hand the arena back, or the learner rebuilds its derived weight copies before every forward from here on
    check(unsafe { ffi::bdr_agent_arena_release(learner.raw(), 0) })?;
    let mut mailbox = std::ptr::null_mut();
    check(unsafe { ffi::bdr_model_mailbox_create(device, n_floats, n_actors as u32, &mut mailbox) })?;
    mailbox_guard.0 = mailbox;

    // environments behind vtables; built lazily on the actor threads
    let mut env_ctxs: Vec<Box<EnvCtx<E, A>>> = (0..n_actors)
        .map(|i| {
            Box::new(EnvCtx::<E, A> {
                env: None,
                config: env_config_train.clone(),
                seed: i as i64, // actor_manager/base.rs:150
                obs_row_bytes: O::ROW_BYTES,
                act_row_bytes,
                _a: std::marker::PhantomData,
            })
        })
        .collect();

    let mut learner_ops: ffi::bdr_learner_ops = unsafe { std::mem::zeroed() };
    unsafe { ffi::bdr_learner_ops_default(&mut learner_ops, learner.raw(), buffer.handle(), mailbox) };
    let xctx = comm.map(|c| ExchangeCtx { comm: c, which: 0 });
    if let Some(x) = &xctx {
        learner_ops.exchange = Some(exchange_cb);
        learner_ops.agree = Some(agree_cb);
        learner_ops.exchange_ctx = x as *const ExchangeCtx as *mut c_void;
    }

    let mut actor_ops: Vec<ffi::bdr_actor_ops> = Vec::with_capacity(n_actors);
    for (i, a) in actors.iter().enumerate() {
        let vt = ffi::bdr_env_vtable {
            ctx: env_ctxs[i].as_mut() as *mut EnvCtx<E, A> as *mut c_void,
            reset: Some(env_reset::<E, A>),
            step_with_reset: Some(env_step::<E, A>),
            obs_on_device: 0, // border's Env trait hands observations over as host values
            device: 0,
        };
        let mut ops: ffi::bdr_actor_ops = unsafe { std::mem::zeroed() };
        unsafe { ffi::bdr_actor_ops_default(&mut ops, a.raw(), mailbox, &vt) };
        actor_ops.push(ops);
    }

    let mut c: ffi::bdr_async_trainer_config = unsafe { std::mem::zeroed() };
    unsafe { ffi::bdr_async_trainer_config_default(&mut c) };
    c.max_opts = async_trainer_config.max_opts as u64;
    c.warmup_period = async_trainer_config.warmup_period as u64;
    c.sync_interval = async_trainer_config.sync_interval as u64;
    c.record_agent_info_interval = async_trainer_config.record_agent_info_interval as u64;
    c.record_compute_cost_interval = async_trainer_config.record_compute_cost_interval as u64;
    c.n_buffer = actor_man_config.n_buffer as u64;
    c.obs_row_bytes = O::ROW_BYTES as u64;
    c.act_row_bytes = act_row_bytes as u64;

    let mut octx = ObserverCtx { keys: record_keys(learner.raw())?, recorder };
    let mut stat = ffi::bdr_async_stats::default();
    let mut astats = vec![ffi::bdr_actor_stat::default(); n_actors];
    let rc = unsafe {
        ffi::bdr_async_train(
            &c,
            &learner_ops,
            actor_ops.as_ptr(),
            n_actors as u32,
            Some(observer_cb),
            &mut octx as *mut ObserverCtx as *mut c_void,
            &mut stat,
            astats.as_mut_ptr(),
        )
    };
    // the loop's own status first; the mailbox is destroyed whatever happened (the actor threads have been joined): by its guard,
    // after the actors' handles - here and on every early return above
    let run = check(rc);
    let wait = run.is_ok().then(|| check(unsafe { ffi::bdr_agent_sync(learner.raw()) }));
    for a in &actors {
        unsafe { ffi::bdr_agent_sync(a.raw()) };
    }
    drop(actors);
    drop(env_ctxs.drain(..));
    drop(mailbox_guard);
    run?;
    if let Some(w) = wait {
        w?;
    }
    learner.eval();

    let out = AmdAsyncTrainStat {
        stat: AsyncTrainStat {
            samples_per_sec: stat.samples_per_sec,
            duration: Duration::from_secs_f64(stat.duration_s),
            opt_per_sec: stat.opt_per_sec,
        },
        actor_stats: astats.iter().map(|s| ActorStat { env_steps: s.env_steps as usize, duration: Duration::from_secs_f64(s.duration_s) }).collect(),
        samples_total: stat.samples_total as usize,
        opt_steps: stat.opt_steps as usize,
        n_syncs: stat.n_syncs as usize,
        n_messages: stat.n_messages as usize,
    };
    drop(buffer);
    Ok((learner, out))
}
