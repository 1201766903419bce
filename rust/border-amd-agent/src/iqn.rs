//! `Iqn` (`border-tch-agent/src/iqn/base.rs`) over the C ABI.
use crate::{
    bytes::{DiscreteAct, ObsRows, RowBatch},
    config::IqnConfig,
    error::expect,
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

/// IQN agent on one MI355X (`Iqn<E, F, M, R>`; feature extractor and merge network are chosen by the config).
pub struct AmdIqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    pub(crate) a: AgentHandle,
    train: bool,
    n_actions: usize,
    phantom: PhantomData<(E, O, A)>,
}

impl<E, O, A> AmdIqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    pub fn handle(&self) -> *mut ffi::bdr_agent {
        self.a.h
    }

    pub fn n_opts(&self) -> usize {
        self.a.n_opts()
    }

    pub fn sync(&mut self) -> Result<()> {
        self.a.sync()
    }

    /// Action values averaged over `sample_percents_act` and the greedy actions (iqn/base.rs:209-215).
    pub fn qvalues(&mut self, obs: &E::Obs) -> (Vec<f32>, Vec<i64>)
    where
        E::Obs: ObsRows,
    {
        let n = obs.n_procs();
        let mut q = vec![0f32; n * self.n_actions];
        let mut best = vec![0i64; n];
        expect(
            unsafe { ffi::bdr_iqn_qvalues(self.a.h, n as u64, obs.as_bytes().as_ptr() as *const c_void, q.as_mut_ptr(), best.as_mut_ptr()) },
            "bdr_iqn_qvalues",
        );
        (q, best)
    }
}

impl<E, O, A> Policy<E> for AmdIqn<E, O, A>
where
    E: Env,
    E::Obs: ObsRows,
    E::Act: DiscreteAct,
    O: RowBatch,
    A: RowBatch,
{
    /// iqn/base.rs:204-228: quantile average over `sample_percents_act`, then `IqnConfig::explorer` (training) / argmax (evaluation).
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

impl<E, O, A> Configurable for AmdIqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    type Config = IqnConfig;

    /// iqn/base.rs:230-268.
    fn build(config: Self::Config) -> Self {
        let c = config.to_c().expect("IqnConfig");
        let mut h = std::ptr::null_mut();
        expect(unsafe { ffi::bdr_iqn_create(&c, &mut h) }, "Iqn::build");
        let e = config.explorer.to_c(config.seed);
        expect(unsafe { ffi::bdr_agent_set_explorer(h, &e) }, "bdr_agent_set_explorer");
        Self { a: AgentHandle::new(h), train: config.train, n_actions: c.n_actions as usize, phantom: PhantomData }
    }
}

impl<E, O, A> Agent<E, AmdReplayBuffer<O, A>> for AmdIqn<E, O, A>
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

    /// iqn/base.rs:172-191 (`opt_`): `n_updates_per_opt` x `update_critic` (:63-170), soft update, `n_opts += 1`.
    fn opt(&mut self, buffer: &mut AmdReplayBuffer<O, A>) {
        self.a.opt(buffer.h);
    }

    /// `loss_critic` (iqn/base.rs:190).
    fn opt_with_record(&mut self, buffer: &mut AmdReplayBuffer<O, A>) -> Record {
        self.a.opt_with_record(buffer.h)
    }

    /// iqn/base.rs:303-311: `iqn.pt.tch`, `iqn_tgt.pt.tch`.
    fn save_params(&self, path: &Path) -> Result<Vec<PathBuf>> {
        self.a.save_params(path, &["iqn".to_string(), "iqn_tgt".to_string()])
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

impl<E, O, A> SyncModel for AmdIqn<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    type ModelInfo = Vec<f32>;

    /// iqn/base.rs:335-356: the whole `iqn` model (feature extractor, cosine embedding, merge network).
    fn model_info(&self) -> (usize, Self::ModelInfo) {
        (self.a.n_opts(), self.a.get_params(0))
    }

    fn sync_model(&mut self, model_info: &Self::ModelInfo) {
        self.a.set_params(0, model_info);
    }
}
