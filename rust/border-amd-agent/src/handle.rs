//! The opaque `bdr_agent` handle and what every agent kind does with it.
use crate::{
    error::{check, expect},
    ffi,
};
use anyhow::{anyhow, Result};
use border_core::record::{Record, RecordValue};
use std::{
    ffi::CString,
    os::raw::c_char,
    path::{Path, PathBuf},
};

pub(crate) struct AgentHandle {
    pub(crate) h: *mut ffi::bdr_agent,
    keys: Vec<String>,
    /// `*.safetensors` instead of the default `*.pt.tch` (set through `set_checkpoint_format`, mirrored here so that
    /// `save_params` can name the files the library actually wrote)
    safetensors: bool,
    /// A device-side condition of an EARLIER step that `Agent::opt` / `opt_with_record` reported (`bdr_last_error_is_deferred`): logged
    /// when it happens and returned by the next call that can return an error (`sync`, `save_params`).  `RefCell`: `save_params` has `&self`.
    deferred: std::cell::RefCell<Option<anyhow::Error>>,
}

// SAFETY: one HIP stream set per handle, hipSetDevice on every entry; `Send`, not `Sync` (all device work behind `&mut`).
unsafe impl Send for AgentHandle {}

impl AgentHandle {
    pub(crate) fn new(h: *mut ffi::bdr_agent) -> Self {
        Self { h, keys: Vec::new(), safetensors: false, deferred: std::cell::RefCell::new(None) }
    }

    pub(crate) fn set_train(&mut self, on: bool) {
        expect(unsafe { ffi::bdr_agent_set_train(self.h, on as i32) }, "bdr_agent_set_train");
    }

    pub(crate) fn is_train(&self) -> bool {
        let mut v = 0i32;
        expect(unsafe { ffi::bdr_agent_is_train(self.h, &mut v) }, "bdr_agent_is_train");
        v != 0
    }

    /// `Agent::opt`: enqueues the step on the agent's streams and returns without waiting for the device.
    ///
    /// Two kinds of failure come back from `bdr_agent_opt`, and `bdr_last_error_is_deferred` tells them apart:
    /// * a device-side condition of an EARLIER step, noticed here without synchronising (an action index outside `[0, n_actions)`, a
    ///   cross-queue wait that timed out, a NaN priority).  The library has cleared it, the state is that of the last good update and
    ///   THIS step has not been enqueued yet: log it, keep it for the next fallible call, and enqueue the step again - `Trainer` counts
    ///   an opt step for every call (`border-core/src/trainer.rs:214-225`), so the call must not return without one.
    /// * a failure of the call itself - an empty buffer (`BDR_ERR_EMPTY`: the reference's `batch()` error is `unwrap()`ed,
    ///   `dqn/base.rs:62`), a buffer whose rows do not fit the network, a wrong device, a HIP error of a launch: the reference panics
    ///   in these places, and so does this (a run must not reach `max_opts` with zero updates and one log line).
    pub(crate) fn opt(&mut self, buffer: *mut ffi::bdr_replay) {
        let mut rc = unsafe { ffi::bdr_agent_opt(self.h, buffer) };
        if rc != ffi::BDR_OK && self.defer_if_async(rc, "Agent::opt") {
            rc = unsafe { ffi::bdr_agent_opt(self.h, buffer) };
        }
        expect(rc, "Agent::opt");
    }

    /// `true` when `rc` reports an earlier step's device-side condition (logged and kept); `false` when it is the call's own failure.
    fn defer_if_async(&mut self, rc: i32, what: &str) -> bool {
        if unsafe { ffi::bdr_last_error_is_deferred() } == 0 {
            return false;
        }
        let e = check(rc).unwrap_err();
        log::error!("{what}: {e:#}");
        self.deferred.borrow_mut().get_or_insert(e);
        true
    }

    fn take_deferred(&self) -> Result<()> {
        match self.deferred.borrow_mut().take() {
            Some(e) => Err(e),
            None => Ok(()),
        }
    }

    /// `VarStore::save`'s other container for the files `save_params` writes (`BDR_CKPT_SAFETENSORS` = 1, `BDR_CKPT_PT_TCH` = 0).
    pub(crate) fn set_checkpoint_format(&mut self, safetensors: bool) -> Result<()> {
        check(unsafe { ffi::bdr_agent_set_checkpoint_format(self.h, safetensors as i32) })?;
        self.safetensors = safetensors;
        Ok(())
    }

    /// Names of the scalars `opt_with_record` returns, in order (the keys of the reference's `Record`).
    fn record_keys(&mut self) -> &[String] {
        if self.keys.is_empty() {
            let mut buf = vec![0u8; 16384];
            let mut n = 0i32;
            expect(
                unsafe { ffi::bdr_agent_record_keys(self.h, buf.as_mut_ptr() as *mut c_char, buf.len() as u64, &mut n) },
                "bdr_agent_record_keys",
            );
            let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
            let text = String::from_utf8_lossy(&buf[..end]).into_owned();
            self.keys = text.split('\n').filter(|s| !s.is_empty()).map(|s| s.to_string()).collect();
            debug_assert_eq!(self.keys.len(), n as usize);
        }
        &self.keys
    }

    /// `Agent::opt_with_record`: the same step, then waits and returns the agent's `Record`
    /// (DQN `dqn/base.rs:311-343`, IQN `iqn/base.rs:285-301`, SAC `sac/base.rs:296-311`).
    pub(crate) fn opt_with_record(&mut self, buffer: *mut ffi::bdr_replay) -> Record {
        let mut vals = vec![0f32; 256];
        let mut n = 0i32;
        // the same policy as `opt`: an earlier step's device-side report is logged and kept, then the step runs.  The library tells the
        // two kinds of report apart (bdr_last_error_is_deferred): 1 = raised by the poll in front of the step, which was therefore never
        // enqueued - run it now; 2 = raised by the check behind the step this call ran - the step is done (n_opts advanced once) and its
        // record is in `vals` / `n`: keep both, do NOT run a second step.
        let mut rc = unsafe { ffi::bdr_agent_opt_with_scalars(self.h, buffer, vals.as_mut_ptr(), vals.len() as i32, &mut n) };
        if rc != ffi::BDR_OK {
            let kind = unsafe { ffi::bdr_last_error_is_deferred() };
            if self.defer_if_async(rc, "Agent::opt_with_record") {
                rc = if kind == 2 {
                    ffi::BDR_OK
                } else {
                    unsafe { ffi::bdr_agent_opt_with_scalars(self.h, buffer, vals.as_mut_ptr(), vals.len() as i32, &mut n) }
                };
                if rc != ffi::BDR_OK && unsafe { ffi::bdr_last_error_is_deferred() } == 2 && self.defer_if_async(rc, "Agent::opt_with_record") {
                    rc = ffi::BDR_OK; // the retried step ran and reported on itself: same rule
                }
            }
        }
        expect(rc, "Agent::opt_with_record");
        let keys = self.record_keys().to_vec();
        let mut record = Record::empty();
        for (k, v) in keys.iter().zip(vals.iter().take(n as usize)) {
            record.insert(k.clone(), RecordValue::Scalar(*v));
        }
        record
    }

    /// The synchronisation's own error first (it is the newer one; a kept report is logged with it), else the kept report.
    pub(crate) fn sync(&mut self) -> Result<()> {
        let now = check(unsafe { ffi::bdr_agent_sync(self.h) });
        let kept = self.take_deferred();
        match (now, kept) {
            (Err(e), Err(k)) => Err(e.context(format!("(an earlier report was pending as well: {k:#})"))),
            (Err(e), Ok(())) => Err(e),
            (Ok(()), kept) => kept,
        }
    }

    pub(crate) fn n_opts(&self) -> usize {
        let mut n = 0u64;
        expect(unsafe { ffi::bdr_agent_n_opts(self.h, &mut n) }, "bdr_agent_n_opts");
        n as usize
    }

    pub(crate) fn param_count_of(&self, which: i32) -> usize {
        let mut n = 0u64;
        expect(unsafe { ffi::bdr_agent_param_count_of(self.h, which, &mut n) }, "bdr_agent_param_count_of");
        n as usize
    }

    /// Parameters of model `which` in the reference's variable order and layouts (`c1.weight` OIHW ... / `mlp.ln{i}.*`).
    pub(crate) fn get_params(&self, which: i32) -> Vec<f32> {
        let n = self.param_count_of(which);
        let mut out = vec![0f32; n];
        expect(unsafe { ffi::bdr_agent_get_params(self.h, which, out.as_mut_ptr(), n as u64) }, "bdr_agent_get_params");
        out
    }

    pub(crate) fn set_params(&mut self, which: i32, p: &[f32]) {
        expect(unsafe { ffi::bdr_agent_set_params(self.h, which, p.as_ptr(), p.len() as u64) }, "bdr_agent_set_params");
    }

    /// `fs::create_dir_all(path)` + the model files under it, in the reference's container (`*.pt.tch`, VarStore::save) or
    /// `*.safetensors` when that format was selected.  `stems`: the file names without extension (`qnet`, `qnet_tgt`, ...); the
    /// returned paths are the files the library wrote.
    pub(crate) fn save_params(&self, path: &Path, stems: &[String]) -> Result<Vec<PathBuf>> {
        let dir = CString::new(path.to_str().ok_or_else(|| anyhow!("non-UTF-8 path"))?)?;
        check(unsafe { ffi::bdr_agent_save_params(self.h, dir.as_ptr()) })?;
        // the files are written (the state of the last good update); a report kept from an earlier `opt` is returned now, as documented
        self.take_deferred()?;
        let ext = if self.safetensors { "safetensors" } else { "pt.tch" };
        Ok(stems.iter().map(|f| path.join(format!("{f}.{ext}"))).collect())
    }

    /// A report kept from an earlier `opt` does not stop a checkpoint from loading (the parameters it concerned are being replaced):
    /// it is logged and dropped.
    pub(crate) fn load_params(&mut self, path: &Path) -> Result<()> {
        if let Err(e) = self.take_deferred() {
            log::warn!("load_params: dropping a pending report of an earlier step: {e:#}");
        }
        let dir = CString::new(path.to_str().ok_or_else(|| anyhow!("non-UTF-8 path"))?)?;
        check(unsafe { ffi::bdr_agent_load_params(self.h, dir.as_ptr()) })
    }
}

impl Drop for AgentHandle {
    fn drop(&mut self) {
        unsafe {
            ffi::bdr_agent_destroy(self.h);
        }
    }
}
