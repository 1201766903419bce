//! `SimpleReplayBuffer` (`border-core/src/generic_replay_buffer/base.rs:86-123`) as a ring in HBM.
//!
//! Same `Config` (`SimpleReplayBufferConfig`), same `Item` / `Batch` (`GenericTransitionBatch<O, A>`), same index stream
//! (`StdRng::seed_from_u64(seed)`, `ix = next_u32() as usize % size`, with replacement, `base.rs:384-390`), same ring
//! arithmetic in `push` (`base.rs:295-316`).  The agents of this crate never call `batch()`: `Agent::opt` hands the buffer's
//! handle to the library, which draws and gathers on the device; `batch()` exists for callers that want the host copy.
use crate::{
    bytes::RowBatch,
    error::{check, expect},
    ffi,
};
use anyhow::Result;
use border_core::{
    generic_replay_buffer::{GenericTransitionBatch, PerConfig, SimpleReplayBufferConfig, WeightNormalizer},
    ExperienceBufferBase, ReplayBufferBase, TransitionBatch,
};
use std::{marker::PhantomData, os::raw::c_void};

/// Where the ring lives and how observations are stored; everything `SimpleReplayBufferConfig` does not say.
#[derive(Clone, Copy, Debug)]
pub struct AmdReplayPlacement {
    /// HIP device ordinal.
    pub device: i32,
    /// `0`: rows stored as pushed.  `k > 0` (Atari: 4): single-frame store - every distinct frame of the k-stacks is kept
    /// once (`border-atari-env/src/env.rs:197-209`), 8x less HBM for k = 4; indices, pushes and batches are unchanged.
    pub frame_stack: i32,
    /// Frames in the store (`0`: `capacity + capacity / 4 + 64`).
    pub frame_capacity: u64,
    /// `false`: the reference's `StdRng::seed_from_u64(seed)` index stream, bit for bit (`base.rs:353, 386`).  `true`: the
    /// device-native xoshiro256++ generator (one per batch lane in HBM) - not the reference's stream, uniform sampling only.
    pub xoshiro_indices: bool,
}

impl Default for AmdReplayPlacement {
    fn default() -> Self {
        Self { device: 0, frame_stack: 0, frame_capacity: 0, xoshiro_indices: false }
    }
}

pub struct AmdReplayBuffer<O, A>
where
    O: RowBatch,
    A: RowBatch,
{
    pub(crate) h: *mut ffi::bdr_replay,
    capacity: usize,
    per: bool,
    phantom: PhantomData<(O, A)>,
}

// SAFETY: the handle owns its HIP stream and buffers; the library sets the device on every entry.  Movable between
// threads, not shareable (`&mut self` on everything that touches the device), exactly like `SimpleReplayBuffer`.
unsafe impl<O: RowBatch, A: RowBatch> Send for AmdReplayBuffer<O, A> {}

fn per_to_c(per: &PerConfig) -> ffi::bdr_per_config {
    ffi::bdr_per_config {
        alpha: per.alpha,
        beta_0: per.beta_0,
        beta_final: per.beta_final,
        n_opts_final: per.n_opts_final as u64,
        normalize: match per.normalize {
            WeightNormalizer::All => ffi::BDR_PER_NORMALIZE_ALL,
            WeightNormalizer::Batch => ffi::BDR_PER_NORMALIZE_BATCH,
        },
        reserved: 0,
    }
}

impl<O, A> AmdReplayBuffer<O, A>
where
    O: RowBatch,
    A: RowBatch,
{
    /// `build` on a chosen device / with the single-frame store.
    pub fn build_on(config: &SimpleReplayBufferConfig, place: AmdReplayPlacement) -> Result<Self> {
        let cfg = ffi::bdr_replay_config {
            capacity: config.capacity as u64,
            seed: config.seed,
            obs_row_bytes: O::ROW_BYTES as u64,
            act_row_bytes: A::ROW_BYTES as u64,
            device: place.device,
            frame_stack: place.frame_stack,
            frame_capacity: place.frame_capacity,
            index_rng: if place.xoshiro_indices { ffi::BDR_RNG_XOSHIRO256PP } else { ffi::BDR_RNG_STDRNG },
            reserved: 0,
        };
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::bdr_replay_create(&cfg, &mut h) })?;
        let mut this = Self { h, capacity: config.capacity, per: false, phantom: PhantomData };
        if let Some(per) = &config.per_config {
            // base.rs:341-345: per_state = Some(PerState::new(capacity, per_config))
            let c = per_to_c(per);
            check(unsafe { ffi::bdr_replay_enable_per(this.h, &c) })?;
            this.per = true;
        }
        Ok(this)
    }

    /// The opaque handle, for the C-side loops (`bdr_trainer_train`, `bdr_async_train`).
    pub fn handle(&self) -> *mut ffi::bdr_replay {
        self.h
    }

    pub fn capacity(&self) -> usize {
        self.capacity
    }

    /// The write cursor `i` of the reference's struct.
    pub fn head(&self) -> usize {
        let mut i = 0u64;
        expect(unsafe { ffi::bdr_replay_head(self.h, &mut i) }, "bdr_replay_head");
        i as usize
    }
}

impl<O, A> ExperienceBufferBase for AmdReplayBuffer<O, A>
where
    O: RowBatch,
    A: RowBatch,
{
    type Item = GenericTransitionBatch<O, A>;

    /// `base.rs:295-316`: rows at `(i + k) % capacity`, `i = (i + len) % capacity`, `size = min(size + len, capacity)`;
    /// with PER every new row gets the current maximum priority (`:227-235`).
    fn push(&mut self, tr: Self::Item) -> Result<()> {
        let n = tr.len();
        let (obs, act, next_obs, reward, is_terminated, is_truncated, _, _) = tr.unpack();
        debug_assert_eq!(obs.n_rows(), n);
        debug_assert_eq!(act.n_rows(), n);
        debug_assert_eq!(next_obs.n_rows(), n);
        check(unsafe {
            ffi::bdr_replay_push(
                self.h,
                n as u64,
                obs.as_bytes().as_ptr() as *const c_void,
                act.as_bytes().as_ptr() as *const c_void,
                next_obs.as_bytes().as_ptr() as *const c_void,
                reward.as_ptr(),
                is_terminated.as_ptr(),
                is_truncated.as_ptr(),
            )
        })
    }

    fn len(&self) -> usize {
        let mut n = 0u64;
        expect(unsafe { ffi::bdr_replay_len(self.h, &mut n) }, "bdr_replay_len");
        n as usize
    }
}

impl<O, A> ReplayBufferBase for AmdReplayBuffer<O, A>
where
    O: RowBatch,
    A: RowBatch,
{
    type Config = SimpleReplayBufferConfig;
    type Batch = GenericTransitionBatch<O, A>;

    /// `base.rs:336-356`.  Device 0, rows stored as pushed; [`AmdReplayBuffer::build_on`] for anything else.
    fn build(config: &Self::Config) -> Self {
        Self::build_on(config, AmdReplayPlacement::default()).expect("AmdReplayBuffer::build")
    }

    /// `base.rs:376-402`: host copy of the batch the device drew and gathered.
    fn batch(&mut self, size: usize) -> Result<Self::Batch> {
        let mut ixs = vec![0u64; size];
        let mut obs = vec![0u8; size * O::ROW_BYTES];
        let mut act = vec![0u8; size * A::ROW_BYTES];
        let mut next_obs = vec![0u8; size * O::ROW_BYTES];
        let mut reward = vec![0f32; size];
        let mut is_terminated = vec![0i8; size];
        let mut is_truncated = vec![0i8; size];
        check(unsafe {
            ffi::bdr_replay_batch(
                self.h,
                size as u64,
                ixs.as_mut_ptr(),
                obs.as_mut_ptr() as *mut c_void,
                act.as_mut_ptr() as *mut c_void,
                next_obs.as_mut_ptr() as *mut c_void,
                reward.as_mut_ptr(),
                is_terminated.as_mut_ptr(),
                is_truncated.as_mut_ptr(),
            )
        })?;
        let weight = if self.per {
            let mut w = vec![0f32; size];
            check(unsafe { ffi::bdr_replay_batch_weights(self.h, size as u64, w.as_mut_ptr()) })?;
            Some(w) // base.rs:377-383
        } else {
            None
        };
        Ok(GenericTransitionBatch {
            obs: O::from_bytes(obs, size),
            act: A::from_bytes(act, size),
            next_obs: O::from_bytes(next_obs, size),
            reward,
            is_terminated,
            is_truncated,
            weight,
            ix_sample: Some(ixs.into_iter().map(|i| i as usize).collect()),
        })
    }

    /// `base.rs:413-426`: no-op without PER (like the reference); with PER `sum_tree.update(ix, td_err)` in order, then the
    /// importance-weight schedule advances by one optimisation step.  This crate's DQN does it on the device after its own
    /// backward (`dqn/base.rs:143`); the method is for agents that live on the Rust side.
    fn update_priority(&mut self, ixs: &Option<Vec<usize>>, td_err: &Option<Vec<f32>>) {
        if !self.per {
            return;
        }
        let ixs = ixs.as_ref().expect("ixs should be Some(_) when per_state is Some(_)");
        let td = td_err.as_ref().expect("td_errs should be Some(_) when per_state is Some(_)");
        let ixs: Vec<u64> = ixs.iter().map(|&i| i as u64).collect();
        expect(unsafe { ffi::bdr_replay_update_priority(self.h, ixs.len() as u64, ixs.as_ptr(), td.as_ptr()) }, "update_priority");
    }
}

impl<O, A> Drop for AmdReplayBuffer<O, A>
where
    O: RowBatch,
    A: RowBatch,
{
    fn drop(&mut self) {
        // Agents remember the buffer of their last opt by uid, not by pointer, so dropping the buffer first is fine.
        unsafe {
            ffi::bdr_replay_destroy(self.h);
        }
    }
}
