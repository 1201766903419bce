//! border's DQN / IQN / SAC agents and `SimpleReplayBuffer` on MI355X.
//!
//! Drop-in for `border-tch-agent` on the opt-step path: the same border-core traits ([`Agent`](border_core::Agent),
//! [`Policy`](border_core::Policy), [`Configurable`](border_core::Configurable),
//! [`ReplayBufferBase`](border_core::ReplayBufferBase), [`ExperienceBufferBase`](border_core::ExperienceBufferBase)) and
//! border-async-trainer's [`SyncModel`](border_async_trainer::SyncModel), implemented over `libborder_amd.so` - hand-written
//! HIP kernels for gfx950 behind the C ABI of `include/border_amd.h`.  No tch, no candle: nothing here links libtorch.
//!
//! * [`AmdReplayBuffer`] - `SimpleReplayBuffer` (`border-core/src/generic_replay_buffer/base.rs`) as a ring in HBM; indices of
//!   `batch()` are those of `StdRng::seed_from_u64(seed)`, bit for bit.
//! * [`AmdDqn`], [`AmdIqn`], [`AmdSac`] - `border-tch-agent/src/{dqn,iqn,sac}/base.rs`; configs keep the reference's field names
//!   and serde layout ([`config`]), so the example YAML files load unchanged.
//! * [`train_async`] - `border-async-trainer/src/util.rs:31-92` on one GPU (learner + actors + device mailbox), with the
//!   optional cross-GPU exchange over RCCL ([`Comm`]).
//!
//! An example binary changes two type aliases (`examples/atari/dqn_atari_tch/src/main.rs:28-45`):
//! `type Agent_ = AmdDqn<Env, ObsBatch, ActBatch>; type ReplayBuffer_ = AmdReplayBuffer<ObsBatch, ActBatch>;`
//! and keeps its `Trainer::build(config).train(env, step_proc, &mut agent, &mut buffer, ...)` call.
pub mod async_trainer;
pub mod bytes;
pub mod comm;
pub mod config;
pub mod dqn;
pub mod error;
pub mod ffi;
mod handle;
pub mod iqn;
pub mod replay;
pub mod sac;

pub use async_trainer::{train_async, AmdAsyncTrainStat};
pub use bytes::{ActFromRows, DiscreteAct, FloatAct, ObsRows, RowBatch};
pub use comm::Comm;
pub use config::{
    ActorConfig, AtariCnnConfig, CriticConfig, CriticLoss, Device, DqnConfig, DqnExplorer, DqnModelConfig, EntCoefMode, EpsilonGreedy,
    Arithmetic, IqnConfig, IqnExplorer, IqnModelConfig, IqnSample, MlpConfig, OptimizerConfig, QNetConfig, SacConfig, Softmax,
};
pub use dqn::AmdDqn;
pub use iqn::AmdIqn;
pub use replay::AmdReplayBuffer;
pub use sac::AmdSac;
