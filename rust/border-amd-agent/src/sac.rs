//! `Sac` (`border-tch-agent/src/sac/base.rs`) over the C ABI.  Actor = `Mlp2`, critics = `Mlp` on `cat(obs, act)`.
use crate::{
    bytes::{FloatAct, ObsRows, RowBatch},
    config::SacConfig,
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
    path::{Path, PathBuf},
};

/// SAC agent on one MI355X (`Sac<E, Q, P, R>`).  Observation rows are `obs_dim` f32, action rows `act_dim` f32.
pub struct AmdSac<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    pub(crate) a: AgentHandle,
    train: bool,
    act_dim: usize,
    n_critics: usize,
    phantom: PhantomData<(E, O, A)>,
}

impl<E, O, A> AmdSac<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    /// Parameter model ids of `bdr_agent_{get,set}_params` for SAC.
    pub const PI: i32 = 0;

    pub fn qnet(&self, i: usize) -> i32 {
        1 + i as i32
    }

    pub fn qnet_tgt(&self, i: usize) -> i32 {
        1 + (self.n_critics + i) as i32
    }

    pub fn log_alpha(&self) -> i32 {
        1 + 2 * self.n_critics as i32
    }

    pub fn handle(&self) -> *mut ffi::bdr_agent {
        self.a.h
    }

    pub fn n_opts(&self) -> usize {
        self.a.n_opts()
    }

    pub fn sync(&mut self) -> Result<()> {
        self.a.sync()
    }
}

fn as_f32(bytes: &[u8]) -> &[f32] {
    debug_assert_eq!(bytes.len() % 4, 0);
    debug_assert_eq!(bytes.as_ptr() as usize % 4, 0);
    // SAFETY: ObsRows of a SAC environment hands out the bytes of an f32 buffer (checked above in debug builds).
    unsafe { std::slice::from_raw_parts(bytes.as_ptr() as *const f32, bytes.len() / 4) }
}

impl<E, O, A> Policy<E> for AmdSac<E, O, A>
where
    E: Env,
    E::Obs: ObsRows,
    E::Act: FloatAct,
    O: RowBatch,
    A: RowBatch,
{
    /// sac/base.rs:215-225: training `tanh(mean + std * z)`, evaluation `tanh(mean)`; `z` from the agent's seeded device stream
    /// (the reference draws it from torch's global generator).
    fn sample(&mut self, obs: &E::Obs) -> E::Act {
        let n = obs.n_procs();
        let mut act = vec![0f32; n * self.act_dim];
        expect(unsafe { ffi::bdr_sac_sample(self.a.h, n as u64, as_f32(obs.as_bytes()).as_ptr(), act.as_mut_ptr()) }, "Policy::sample");
        E::Act::from_rows(act, n)
    }
}

impl<E, O, A> Configurable for AmdSac<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    type Config = SacConfig;

    /// sac/base.rs:237-285.
    fn build(config: Self::Config) -> Self {
        let c = config.to_c().expect("SacConfig");
        let mut h = std::ptr::null_mut();
        expect(unsafe { ffi::bdr_sac_create(&c, &mut h) }, "Sac::build");
        Self { a: AgentHandle::new(h), train: config.train, act_dim: c.act_dim as usize, n_critics: c.n_critics as usize, phantom: PhantomData }
    }
}

impl<E, O, A> Agent<E, AmdReplayBuffer<O, A>> for AmdSac<E, O, A>
where
    E: Env + 'static,
    E::Obs: ObsRows,
    E::Act: FloatAct,
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

    /// sac/base.rs:175-198 (`opt_`), per update and in this order: batch; actor (+ entropy coefficient) first (:181); critics
    /// against the UPDATED actor (:182); `track` of every critic (:183).
    fn opt(&mut self, buffer: &mut AmdReplayBuffer<O, A>) {
        self.a.opt(buffer.h);
    }

    /// `loss_critic`, `loss_actor`, `ent_coef` (sac/base.rs:187-196).
    fn opt_with_record(&mut self, buffer: &mut AmdReplayBuffer<O, A>) -> Record {
        self.a.opt_with_record(buffer.h)
    }

    /// sac/base.rs:313-334: `qnet_{i}.pt.tch`, `qnet_tgt_{i}.pt.tch` per critic, then `pi.pt.tch`, `ent_coef.pt.tch`.
    fn save_params(&self, path: &Path) -> Result<Vec<PathBuf>> {
        let mut files = Vec::new();
        for i in 0..self.n_critics {
            files.push(format!("qnet_{}", i));
            files.push(format!("qnet_tgt_{}", i));
        }
        files.push("pi".to_string());
        files.push("ent_coef".to_string());
        self.a.save_params(path, &files)
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

impl<E, O, A> SyncModel for AmdSac<E, O, A>
where
    E: Env,
    O: RowBatch,
    A: RowBatch,
{
    type ModelInfo = Vec<f32>;

    /// sac/base.rs:377-386: actors only need `pi`.
    fn model_info(&self) -> (usize, Self::ModelInfo) {
        (self.a.n_opts(), self.a.get_params(Self::PI))
    }

    fn sync_model(&mut self, model_info: &Self::ModelInfo) {
        self.a.set_params(Self::PI, model_info);
    }
}
