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
    /// A device-side failure reported by `Agent::opt` (which returns `()` in the reference's trait): logged when it happens and
    /// returned by the next call that can return an error (`sync`, `save_params`, `load_params`).
    deferred: Option<anyhow::Error>,
}

// SAFETY: one HIP stream set per handle, hipSetDevice on every entry; `Send`, not `Sync` (all device work behind `&mut`).
unsafe impl Send for AgentHandle {}

impl AgentHandle {
    pub(crate) fn new(h: *mut ffi::bdr_agent) -> Self {
        Self { h, keys: Vec::new(), safetensors: false, deferred: None }
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
    /// The library reports device-side conditions of EARLIER steps here without synchronising (an action index outside
    /// `[0, n_actions)`, a cross-queue wait that timed out).  They must not panic the learner thread: the state stays that of the
    /// last good update and training can go on, so the error is logged and kept for the next fallible call.
    pub(crate) fn opt(&mut self, buffer: *mut ffi::bdr_replay) {
        if let Err(e) = check(unsafe { ffi::bdr_agent_opt(self.h, buffer) }) {
            log::error!("Agent::opt: {e:#}");
            self.deferred.get_or_insert(e);
        }
    }

    fn take_deferred(&mut self) -> Result<()> {
        match self.deferred.take() {
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
        expect(
            unsafe { ffi::bdr_agent_opt_with_scalars(self.h, buffer, vals.as_mut_ptr(), vals.len() as i32, &mut n) },
            "Agent::opt_with_record",
        );
        let keys = self.record_keys().to_vec();
        let mut record = Record::empty();
        for (k, v) in keys.iter().zip(vals.iter().take(n as usize)) {
            record.insert(k.clone(), RecordValue::Scalar(*v));
        }
        record
    }

    pub(crate) fn sync(&mut self) -> Result<()> {
        let now = check(unsafe { ffi::bdr_agent_sync(self.h) });
        self.take_deferred()?;
        now
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
        let ext = if self.safetensors { "safetensors" } else { "pt.tch" };
        Ok(stems.iter().map(|f| path.join(format!("{f}.{ext}"))).collect())
    }

    pub(crate) fn load_params(&mut self, path: &Path) -> Result<()> {
        self.take_deferred()?;
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
