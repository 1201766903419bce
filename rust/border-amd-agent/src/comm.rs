//! Multi-GPU parameter exchange: one process per GPU, the library's own RCCL communicator over xGMI.  Replaces the learner ->
//! actors `NamedTensors` channel of `border-async-trainer/src/async_trainer/base.rs:268-272` ACROSS GPUs; inside one GPU the
//! channel is the device mailbox (`async_trainer.rs`).
use crate::{error::check, ffi};
use anyhow::Result;

pub struct Comm {
    pub(crate) c: *mut ffi::bdr_comm,
    pub n_ranks: usize,
    pub rank: usize,
}

unsafe impl Send for Comm {}

impl Comm {
    /// Rank 0 creates the id and ships the 128 bytes to the other ranks over any side channel (a file, a socket, MPI).
    pub fn unique_id() -> Result<[u8; ffi::BDR_UNIQUE_ID_BYTES]> {
        let mut id = [0u8; ffi::BDR_UNIQUE_ID_BYTES];
        check(unsafe { ffi::bdr_comm_get_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    /// Collective: every rank calls it with the same id.  Set `GPU_MAX_HW_QUEUES=8` (or `BDR_NRANKS`) before the process's
    /// first HIP call: the overlapped exchange needs five hardware queues (the library asks for them itself when it is loaded
    /// first in a process whose environment names more than one rank).
    pub fn init_rank(id: &[u8; ffi::BDR_UNIQUE_ID_BYTES], n_ranks: usize, rank: usize, device: i32) -> Result<Self> {
        let mut c = std::ptr::null_mut();
        check(unsafe { ffi::bdr_comm_init_rank(id.as_ptr(), n_ranks as i32, rank as i32, device, &mut c) })?;
        Ok(Self { c, n_ranks, rank })
    }

    /// MIN over ranks of `local_ok`: the agreement before a collective (`bdr_learner_ops::agree`).
    pub fn agree(&self, local_ok: bool) -> Result<bool> {
        let mut all = 0i32;
        check(unsafe { ffi::bdr_comm_agree(self.c, local_ok as i32, &mut all) })?;
        Ok(all != 0)
    }

    /// `params <- mean over ranks` of model `which` of `agent` (north_star: periodic `ncclAllReduce` of the Q-net parameters);
    /// asynchronous, on the agent's own queues (per segment beside the backward for the Nature-CNN agent).
    pub fn allreduce_params(&self, agent: *mut ffi::bdr_agent, which: i32) -> Result<()> {
        check(unsafe { ffi::bdr_agent_allreduce_params(agent, self.c, which) })
    }

    /// `params <- root's`: the reference's learner -> actors semantics across ranks.
    pub fn broadcast_params(&self, agent: *mut ffi::bdr_agent, which: i32, root: usize) -> Result<()> {
        check(unsafe { ffi::bdr_agent_broadcast_params(agent, self.c, which, root as i32) })
    }

    /// Lock-step data parallelism: from now on every `Agent::opt` of `agent` all-reduces its gradients before the optimizer step,
    /// so N ranks with batch B/N take exactly the step one rank takes on B rows.  `None`: back to independent steps.
    pub fn set_grad_comm(agent: *mut ffi::bdr_agent, comm: Option<&Comm>) -> Result<()> {
        check(unsafe { ffi::bdr_agent_set_grad_comm(agent, comm.map(|c| c.c).unwrap_or(std::ptr::null_mut())) })
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe {
            ffi::bdr_comm_destroy(self.c);
        }
    }
}
