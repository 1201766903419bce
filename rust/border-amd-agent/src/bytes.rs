//! What crosses the C ABI is rows of bytes.  These traits say how border's batch / observation / action types turn into rows
//! and back; they replace the `Into<Tensor>` / `From<Tensor>` bounds border-tch-agent puts on the same types
//! (`border-tch-agent/src/dqn/base.rs:204-209`, `tensor_batch.rs:20-60`).
use border_core::generic_replay_buffer::BatchBase;

/// A batch type of the replay buffer (`O` / `A` of `GenericTransitionBatch<O, A>`,
/// `border-core/src/generic_replay_buffer/batch.rs:45-71`) whose rows are fixed-size byte strings.
///
/// Atari observations should be the **u8** batch (28 224 bytes per row), not the f32 tensor of
/// `border-atari-env/src/obs.rs:45-52`: the values are integers 0..=255, u8 is lossless and 4x smaller in HBM.
pub trait RowBatch: BatchBase {
    /// Bytes of one row (obs: `4*1*84*84` for Atari; discrete act: 8, one i64; SAC act: `4 * act_dim`).
    const ROW_BYTES: usize;
    /// Number of rows held.
    fn n_rows(&self) -> usize;
    /// All rows, contiguous, row-major.
    fn as_bytes(&self) -> &[u8];
    /// The inverse: `bytes.len() == n_rows * ROW_BYTES`.
    fn from_bytes(bytes: Vec<u8>, n_rows: usize) -> Self;
}

/// An environment observation (`E::Obs`) as `n_procs` rows, the input of `Policy::sample`.
pub trait ObsRows {
    fn n_procs(&self) -> usize;
    fn as_bytes(&self) -> &[u8];
}

/// An environment action (`E::Act`) built from what `Policy::sample` computed.
pub trait ActFromRows<T> {
    fn from_rows(rows: Vec<T>, n_procs: usize) -> Self;
}

/// Marker: discrete actions (one i64 per process; `dqn/base.rs:211-242`, `iqn/base.rs:204-228`).
pub trait DiscreteAct: ActFromRows<i64> {}
impl<T: ActFromRows<i64>> DiscreteAct for T {}

/// Marker: continuous actions (`act_dim` f32 per process; `sac/base.rs:215-225`).
pub trait FloatAct: ActFromRows<f32> {}
impl<T: ActFromRows<f32>> FloatAct for T {}
