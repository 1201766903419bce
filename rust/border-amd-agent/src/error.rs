//! Status codes of the C ABI -> `anyhow::Error` (or the reference's panics where the reference panics).
use crate::ffi;
use anyhow::{anyhow, Result};
use std::ffi::CStr;

/// The library's thread-local message for the last failure (`bdr_last_error`).
pub fn last_error() -> String {
    // SAFETY: bdr_last_error returns a NUL-terminated thread-local buffer that stays valid until the next failing call on
    // this thread; it is copied before anything else is called.
    unsafe {
        let p = ffi::bdr_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// `0` -> `Ok(())`, anything else -> the library's message with the status name in front.
pub fn check(rc: i32) -> Result<()> {
    if rc == ffi::BDR_OK {
        return Ok(());
    }
    let kind = match rc {
        ffi::BDR_ERR_INVALID => "invalid argument",
        ffi::BDR_ERR_NO_DEVICE => "no HIP device",
        ffi::BDR_ERR_HIP => "HIP runtime",
        ffi::BDR_ERR_EMPTY => "empty replay buffer",
        ffi::BDR_ERR_IO => "io",
        ffi::BDR_ERR_COMM => "RCCL",
        _ => "unknown status",
    };
    Err(anyhow!("border_amd ({}): {}", kind, last_error()))
}

/// For the places where border-tch-agent itself `unwrap()`s / `expect()`s (`dqn/base.rs:62`, `:256-259`, ...).
pub fn expect(rc: i32, what: &str) {
    if let Err(e) = check(rc) {
        panic!("{}: {}", what, e);
    }
}
