//! `Dqn` (`border-tch-agent/src/dqn/base.rs`) over the C ABI.
use crate::{
    bytes::{DiscreteAct, ObsRows, RowBatch},
    config::DqnConfig,
    error::{check, expect},
    ffi,
    handle::AgentHandle,
    replay::AmdReplayBuffer,
};
use anyhow::Result;
use border_async_trainer::SyncModel;
use border_core::{record::Record, Agent, Configurable, Env, Policy};
use std::{
    any::Any,
    marker::PhantomData,
    os::raw::c_void,
    path::{Path, PathBuf},
};

/// DQN agent on one MI355X.  `E`: the environment; `O`, `A`: the batch types of its replay buffer
/// (`AmdReplayBuffer<O, A>`), as in `Dqn<E, Q, R>` - the Q-network is chosen by the config (`QNetConfig`), not by a type.
pub struct AmdDqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    pub(crate) a: AgentHandle,
    train: bool,
    config: DqnConfig,
    phantom: PhantomData<(E, O, A)>,
}

impl<E, O, A> AmdDqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    /// The opaque handle, for the C-side loops (`bdr_trainer_train`, `bdr_async_train`).
    pub fn handle(&self) -> *mut ffi::bdr_agent {
        self.a.h
    }

    /// Number of `opt` calls so far (`n_opts` of the reference's struct).
    pub fn n_opts(&self) -> usize {
        self.a.n_opts()
    }

    /// Blocks until the device has finished everything enqueued by `opt`; device-side errors surface here.
    pub fn sync(&mut self) -> Result<()> {
        self.a.sync()
    }

    /// Q(obs) and the greedy actions for `n_procs` observation rows (`qnet.forward`, dqn/base.rs:213).
    pub fn qvalues(&mut self, obs: &E::Obs) -> (Vec<f32>, Vec<i64>)
    where
        E::Obs: ObsRows,
    {
        let n = obs.n_procs();
        let n_act = self.config.model_config.q_config.as_ref().map(|q| q.get_out_dim()).unwrap_or(0) as usize;
        let mut q = vec![0f32; n * n_act];
        let mut best = vec![0i64; n];
        expect(
            unsafe { ffi::bdr_agent_qvalues(self.a.h, n as u64, obs.as_bytes().as_ptr() as *const c_void, q.as_mut_ptr(), best.as_mut_ptr()) },
            "bdr_agent_qvalues",
        );
        (q, best)
    }
}

impl<E, O, A> Policy<E> for AmdDqn<E, O, A>
where
    E: Env,
    E::Obs: ObsRows,
    E::Act: DiscreteAct,
    O: RowBatch,
    A: RowBatch,
{
    /// dqn/base.rs:211-242 in one call: forward on the GPU, then `DqnConfig::explorer` in training mode (epsilon-greedy with one
    /// coin per call and epsilon linear in calls, or softmax-multinomial) / argmax with 1 % random actions in evaluation mode.
    fn sample(&mut self, obs: &E::Obs) -> E::Act {
        let n = obs.n_procs();
        let mut act = vec![0i64; n];
        expect(
            unsafe {
                ffi::bdr_agent_sample(self.a.h, n as u64, obs.as_bytes().as_ptr() as *const c_void, act.as_mut_ptr(), std::ptr::null_mut())
            },
            "Policy::sample",
        );
        E::Act::from_rows(act, n)
    }
}

impl<E, O, A> Configurable for AmdDqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    type Config = DqnConfig;

    /// dqn/base.rs:252-286.  Panics where the reference panics ("No device is given for DQN agent", "q_config is not set.").
    fn build(config: Self::Config) -> Self {
        let c = config.to_c().expect("DqnConfig");
        let mut h = std::ptr::null_mut();
        expect(unsafe { ffi::bdr_dqn_create(&c, &mut h) }, "Dqn::build");
        let e = config.explorer.to_c(config.seed);
        expect(unsafe { ffi::bdr_agent_set_explorer(h, &e) }, "bdr_agent_set_explorer");
        Self { a: AgentHandle::new(h), train: config.train, config, phantom: PhantomData }
    }
}

impl<E, O, A> Agent<E, AmdReplayBuffer<O, A>> for AmdDqn<E, O, A>
where
    E: Env + 'static,
    E::Obs: ObsRows,
    E::Act: DiscreteAct,
    O: RowBatch + 'static,
    A: RowBatch + 'static,
{
    fn train(&mut self) {
        self.train = true;
        self.a.set_train(true);
    }

    fn eval(&mut self) {
        self.train = false;
        self.a.set_train(false);
    }

    fn is_train(&self) -> bool {
        self.train
    }

    /// dqn/base.rs:301-309 -> `opt_` (:182-200): `n_updates_per_opt` x `update_critic`, soft-update bookkeeping, `n_opts += 1`.
    /// Only enqueues: the trainer's next env step overlaps the device.
    fn opt(&mut self, buffer: &mut AmdReplayBuffer<O, A>) {
        self.a.opt(buffer.h);
    }

    /// dqn/base.rs:311-343: `loss`; with `record_verbose_level >= 2` also `pred_mean`, `reward_mean`, `tgt_mean`,
    /// `tgt_minus_pred_mean`, `qnet.param_stats()` and `ratio_best_act` (whose counters it resets).
    fn opt_with_record(&mut self, buffer: &mut AmdReplayBuffer<O, A>) -> Record {
        self.a.opt_with_record(buffer.h)
    }

    /// dqn/base.rs:345-356: `qnet.pt.tch`, `qnet_tgt.pt.tch` in the container tch's `VarStore::save` writes for these names.
    fn save_params(&self, path: &Path) -> Result<Vec<PathBuf>> {
        self.a.save_params(path, &["qnet".to_string(), "qnet_tgt".to_string()])
    }

    fn load_params(&mut self, path: &Path) -> Result<()> {
        self.a.load_params(path)
    }

    fn as_any_ref(&self) -> &dyn Any {
        self
    }

    fn as_any_mut(&mut self) -> &mut dyn Any {
        self
    }
}

impl<E, O, A> SyncModel for AmdDqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    /// `NamedTensors` of the reference (`util/named_tensors.rs`) as one flat vector in the reference's variable order.
    type ModelInfo = Vec<f32>;

    /// dqn/base.rs:390-395.
    fn model_info(&self) -> (usize, Self::ModelInfo) {
        (self.a.n_opts(), self.a.get_params(0))
    }

    /// dqn/base.rs:397-400.
    fn sync_model(&mut self, model_info: &Self::ModelInfo) {
        self.a.set_params(0, model_info);
    }
}

/// `Agent::opt` on a caller-supplied minibatch (parity tests: "fixed minibatch"); returns the loss.
pub fn update_on_batch<E, O, A>(agent: &mut AmdDqn<E, O, A>, obs: &O, act: &[i64], next_obs: &O, reward: &[f32], is_terminated: &[i8]) -> Result<f32>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    let mut rec = ffi::bdr_dqn_record::default();
    check(unsafe {
        ffi::bdr_dqn_update_on_batch(
            agent.a.h,
            reward.len() as u64,
            obs.as_bytes().as_ptr() as *const c_void,
            act.as_ptr(),
            next_obs.as_bytes().as_ptr() as *const c_void,
            reward.as_ptr(),
            is_terminated.as_ptr(),
            &mut rec,
        )
    })?;
    Ok(rec.loss)
}
