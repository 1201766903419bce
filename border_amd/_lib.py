"""ctypes loader for libborder_amd.so (the C ABI declared in include/border_amd.h).

The product path has no CPU fallback: if the HIP extension is missing or no MI355X is visible,
constructors raise.  `import torch` happens first on purpose: PyTorch-ROCm bundles its own
libamdhip64.so.7 / librccl.so.1, and loading it first makes this library bind to that same
runtime by soname instead of pulling a second HIP runtime into the process.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BORDER_AMD_LIB") or os.path.join(HERE, "libborder_amd.so")  # override: kernel A/B probes

BDR_MAX_UNITS = 8
BDR_UNIQUE_ID_BYTES = 128


class BdrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"border_amd error {code}: {msg}")
        self.code = code


class ReplayConfig(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("seed", C.c_uint64), ("obs_row_bytes", C.c_uint64),
                ("act_row_bytes", C.c_uint64), ("device", C.c_int32), ("frame_stack", C.c_int32), ("frame_capacity", C.c_uint64),
                ("index_rng", C.c_int32), ("reserved", C.c_int32)]


class NetConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_stack", C.c_int32), ("in_dim", C.c_int32), ("n_units", C.c_int32),
                ("units", C.c_int32 * BDR_MAX_UNITS), ("out_dim", C.c_int32), ("activation_out", C.c_int32)]


class AdamWConfigC(C.Structure):
    _fields_ = [("opt_kind", C.c_int32), ("amsgrad", C.c_int32), ("beta1", C.c_double), ("beta2", C.c_double),
                ("weight_decay", C.c_double), ("eps", C.c_double)]

    def fill(self, o) -> None:
        """o: border_amd.OptimizerConfig (opt.rs:13-28); lr travels in the owning struct"""
        self.opt_kind = {"Adam": 0, "AdamW": 1}[o.kind]
        self.beta1, self.beta2, self.weight_decay, self.eps = o.beta1, o.beta2, o.wd, o.eps
        self.amsgrad = 1 if (o.kind == "AdamW" and o.amsgrad) else 0


ARITHMETIC = {"bf16x3_6": 0, "f32_exact": 1}   # BDR_ARITH_*


class DqnConfigC(C.Structure):
    _fields_ = [("net", NetConfig), ("opt_kind", C.c_int32), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("weight_decay", C.c_double), ("eps", C.c_double), ("amsgrad", C.c_int32),
                ("soft_update_interval", C.c_uint64), ("n_updates_per_opt", C.c_uint64),
                ("batch_size", C.c_uint64), ("discount_factor", C.c_double), ("tau", C.c_double),
                ("train", C.c_int32), ("double_dqn", C.c_int32), ("critic_loss", C.c_int32),
                ("has_clip_td_err", C.c_int32), ("clip_td_err_min", C.c_double), ("clip_td_err_max", C.c_double),
                ("record_verbose_level", C.c_int32), ("device", C.c_int32), ("param_seed", C.c_uint64),
                ("arithmetic", C.c_int32), ("reserved", C.c_int32)]


class IqnConfigC(C.Structure):
    _fields_ = [("psi", NetConfig), ("feature_dim", C.c_int32), ("embed_dim", C.c_int32), ("n_f_units", C.c_int32),
                ("f_units", C.c_int32 * BDR_MAX_UNITS), ("n_actions", C.c_int32), ("lr", C.c_double),
                ("soft_update_interval", C.c_uint64), ("n_updates_per_opt", C.c_uint64), ("batch_size", C.c_uint64),
                ("discount_factor", C.c_double), ("tau", C.c_double), ("sample_percents_pred", C.c_int32),
                ("sample_percents_tgt", C.c_int32), ("sample_percents_act", C.c_int32), ("train", C.c_int32),
                ("device", C.c_int32), ("seed", C.c_uint64), ("opt", AdamWConfigC), ("arithmetic", C.c_int32), ("reserved", C.c_int32)]


class SacConfigC(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_pi_units", C.c_int32), ("pi_units", C.c_int32 * BDR_MAX_UNITS),
                ("n_q_units", C.c_int32), ("q_units", C.c_int32 * BDR_MAX_UNITS), ("lr_actor", C.c_double), ("lr_critic", C.c_double),
                ("gamma", C.c_double), ("tau", C.c_double), ("ent_coef_auto", C.c_int32), ("ent_coef_alpha", C.c_double),
                ("target_entropy", C.c_double), ("ent_coef_lr", C.c_double), ("epsilon", C.c_double), ("min_lstd", C.c_double),
                ("max_lstd", C.c_double), ("n_updates_per_opt", C.c_uint64), ("batch_size", C.c_uint64), ("train", C.c_int32),
                ("critic_loss", C.c_int32), ("reward_scale", C.c_double), ("n_critics", C.c_int32), ("device", C.c_int32),
                ("seed", C.c_uint64), ("opt_actor", AdamWConfigC), ("opt_critic", AdamWConfigC)]


class DqnRecordC(C.Structure):
    _fields_ = [("loss", C.c_float), ("pred_mean", C.c_float), ("reward_mean", C.c_float),
                ("tgt_mean", C.c_float), ("tgt_minus_pred_mean", C.c_float), ("has_verbose", C.c_int32)]


class PerConfigC(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta_0", C.c_float), ("beta_final", C.c_float), ("n_opts_final", C.c_uint64),
                ("normalize", C.c_int32), ("reserved", C.c_int32)]


class PerInfoC(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_opts", C.c_uint64), ("beta", C.c_float), ("total", C.c_float),
                ("max_p", C.c_float), ("min_p", C.c_float)]


class ExplorerConfigC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("eps_start", C.c_double), ("eps_final", C.c_double),
                ("final_step", C.c_uint64), ("n_calls", C.c_uint64), ("seed", C.c_uint64)]


class SampleInfoC(C.Structure):
    _fields_ = [("eps", C.c_double), ("is_random", C.c_int32), ("n_samples_act", C.c_uint64),
                ("n_samples_best_act", C.c_uint64)]


class NamedTensorC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dims", C.POINTER(C.c_uint64)), ("ndim", C.c_uint32)]


ENV_RESET_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
ENV_STEP_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.c_void_p)
SET_TRAIN_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)
SAMPLE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p)
OPT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
OPT_REC_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32))
PUSH_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_int8))
OBSERVER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_float), C.c_int32)


SAMPLE_DEV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p)
PUSH_DEV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_int8),
                          C.POINTER(C.c_int8))


class EnvVtable(C.Structure):
    # obs_on_device != 0: obs_out / init_obs_out of the callbacks are device buffers on GPU `device` (device-resident observations)
    _fields_ = [("ctx", C.c_void_p), ("reset", ENV_RESET_FN), ("step_with_reset", ENV_STEP_FN), ("obs_on_device", C.c_int32), ("device", C.c_int32)]


class TrainerOps(C.Structure):
    _fields_ = [("agent", C.c_void_p), ("buffer", C.c_void_p), ("agent_set_train", SET_TRAIN_FN), ("agent_sample", SAMPLE_FN),
                ("agent_opt", OPT_FN), ("agent_opt_with_record", OPT_REC_FN), ("buffer_push", PUSH_FN),
                ("agent_sample_device", SAMPLE_DEV_FN), ("buffer_push_device", PUSH_DEV_FN)]


class TrainerConfigC(C.Structure):
    _fields_ = [("max_opts", C.c_uint64), ("opt_interval", C.c_uint64), ("warmup_period", C.c_uint64),
                ("record_agent_info_interval", C.c_uint64), ("record_compute_cost_interval", C.c_uint64),
                ("obs_row_bytes", C.c_uint64), ("act_row_bytes", C.c_uint64)]


class TrainerStatsC(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("opt_steps", C.c_uint64), ("n_records", C.c_uint64), ("n_episodes", C.c_uint64),
                ("opt_seconds", C.c_double), ("sample_seconds", C.c_double)]


class DeviceBatch(C.Structure):
    _fields_ = [("n", C.c_uint64), ("obs", C.c_void_p), ("next_obs", C.c_void_p), ("act", C.c_void_p),
                ("reward", C.c_void_p), ("is_terminated", C.c_void_p), ("is_truncated", C.c_void_p),
                ("ixs", C.c_void_p), ("weight", C.c_void_p)]


# every symbol include/border_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "bdr_last_error", "bdr_last_error_is_deferred", "bdr_device_count", "bdr_version",
    "bdr_replay_create", "bdr_replay_destroy", "bdr_replay_push", "bdr_replay_push_device", "bdr_replay_len", "bdr_replay_head", "bdr_replay_frames_used",
    "bdr_replay_sample_indices", "bdr_replay_batch", "bdr_replay_last_batch", "bdr_replay_fill_synthetic",
    "bdr_replay_read_rows",
    "bdr_per_config_default", "bdr_replay_enable_per", "bdr_replay_update_priority", "bdr_replay_batch_weights",
    "bdr_replay_per_info", "bdr_replay_per_read", "bdr_replay_per_get", "bdr_dqn_update_on_batch_weighted",
    "bdr_dqn_config_default", "bdr_dqn_create", "bdr_agent_destroy", "bdr_agent_set_train", "bdr_agent_is_train",
    "bdr_agent_opt", "bdr_agent_opt_with_record", "bdr_agent_opt_with_scalars", "bdr_agent_record_keys", "bdr_agent_draw_noise", "bdr_agent_param_count_of", "bdr_dqn_update_on_batch", "bdr_agent_qvalues",
    "bdr_explorer_config_default", "bdr_agent_set_explorer", "bdr_agent_get_explorer", "bdr_agent_sample", "bdr_agent_sample_device", "bdr_agent_qvalues_device",
    "bdr_agent_sync", "bdr_agent_n_opts", "bdr_agent_param_count", "bdr_agent_get_params",
    "bdr_agent_set_params", "bdr_agent_arena_device_ptr", "bdr_agent_arena_release", "bdr_agent_save_params", "bdr_agent_load_params", "bdr_agent_set_checkpoint_format",
    "bdr_checkpoint_write", "bdr_checkpoint_read", "bdr_dqn_probe",
    "bdr_agent_profile_enable", "bdr_agent_profile_read",
    "bdr_iqn_config_default", "bdr_iqn_create", "bdr_iqn_update_on_batch", "bdr_iqn_forward", "bdr_iqn_qvalues",
    "bdr_sac_config_default", "bdr_sac_create", "bdr_sac_update_on_batch", "bdr_sac_sample", "bdr_sac_sample_device",
    "bdr_comm_get_unique_id", "bdr_comm_init_rank", "bdr_comm_destroy", "bdr_comm_agree", "bdr_sac_probe", "bdr_agent_allreduce_params",
    "bdr_agent_broadcast_params", "bdr_agent_set_grad_comm", "bdr_dqn_grads_on_batch", "bdr_agent_apply_grads",
    "bdr_atari_prep_create", "bdr_atari_prep_destroy", "bdr_atari_prep_reset", "bdr_atari_prep_step", "bdr_atari_prep_obs",
    "bdr_atari_prep_device_stacks", "bdr_atari_prep_device_prev_stacks", "bdr_atari_prep_copy_stack", "bdr_atari_clip_reward",
    "bdr_trainer_config_default", "bdr_trainer_ops_default", "bdr_trainer_train", "bdr_trainer_train_offline",
    "bdr_model_mailbox_create", "bdr_model_mailbox_destroy", "bdr_agent_publish_model", "bdr_agent_sync_model_from",
    "bdr_async_trainer_config_default", "bdr_learner_ops_default", "bdr_actor_ops_default", "bdr_async_train",
]

_lib = None


def lib() -> C.CDLL:
    """Load the HIP extension; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BdrError(2, f"{LIB_PATH} is missing: run `python -m border_amd.build` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (see module docstring: one HIP runtime per process)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.bdr_last_error.restype = C.c_char_p
    L.bdr_version.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)  # AttributeError here == ABI drift
        if name not in ("bdr_last_error", "bdr_version", "bdr_dqn_config_default", "bdr_sac_config_default", "bdr_iqn_config_default",
                        "bdr_explorer_config_default", "bdr_per_config_default", "bdr_atari_clip_reward", "bdr_trainer_config_default",
                        "bdr_trainer_ops_default", "bdr_async_trainer_config_default", "bdr_learner_ops_default", "bdr_actor_ops_default"):
            fn.restype = C.c_int32
    L.bdr_trainer_config_default.restype = None
    L.bdr_trainer_ops_default.restype = None
    L.bdr_trainer_ops_default.argtypes = [C.POINTER(TrainerOps), C.c_void_p, C.c_void_p]
    L.bdr_trainer_train.argtypes = [C.POINTER(TrainerConfigC), C.POINTER(TrainerOps), C.POINTER(EnvVtable), OBSERVER_FN, C.c_void_p,
                                    C.POINTER(TrainerStatsC)]
    L.bdr_trainer_train_offline.argtypes = [C.POINTER(TrainerConfigC), C.POINTER(TrainerOps), OBSERVER_FN, C.c_void_p,
                                            C.POINTER(TrainerStatsC)]
    L.bdr_atari_clip_reward.restype = C.c_float
    L.bdr_atari_clip_reward.argtypes = [C.c_float, C.c_int32]
    L.bdr_dqn_config_default.restype = None
    L.bdr_sac_config_default.restype = None
    L.bdr_iqn_config_default.restype = None
    L.bdr_explorer_config_default.restype = None
    L.bdr_per_config_default.restype = None
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    L.bdr_replay_create.argtypes = [C.POINTER(ReplayConfig), C.POINTER(vp)]
    L.bdr_replay_destroy.argtypes = [vp]
    L.bdr_replay_push.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
    L.bdr_replay_push_device.argtypes = [vp, u64, vp, u64, vp, vp, u64, vp, vp, vp]
    L.bdr_replay_len.argtypes = [vp, C.POINTER(u64)]
    L.bdr_replay_head.argtypes = [vp, C.POINTER(u64)]
    L.bdr_replay_frames_used.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bdr_replay_sample_indices.argtypes = [vp, u64, vp]
    L.bdr_replay_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp]
    L.bdr_replay_last_batch.argtypes = [vp, C.POINTER(DeviceBatch)]
    L.bdr_replay_fill_synthetic.argtypes = [vp, u64, u64, i32, i32]
    L.bdr_replay_read_rows.argtypes = [vp, u64, u64, vp, vp, vp, vp, vp, vp]
    L.bdr_dqn_config_default.argtypes = [C.POINTER(DqnConfigC)]
    L.bdr_dqn_create.argtypes = [C.POINTER(DqnConfigC), C.POINTER(vp)]
    L.bdr_agent_destroy.argtypes = [vp]
    L.bdr_agent_set_train.argtypes = [vp, i32]
    L.bdr_agent_is_train.argtypes = [vp, C.POINTER(i32)]
    L.bdr_agent_opt.argtypes = [vp, vp]
    L.bdr_agent_opt_with_record.argtypes = [vp, vp, C.POINTER(DqnRecordC)]
    L.bdr_agent_opt_with_scalars.argtypes = [vp, vp, vp, i32, C.POINTER(i32)]
    L.bdr_agent_record_keys.argtypes = [vp, vp, u64, C.POINTER(i32)]
    L.bdr_agent_draw_noise.argtypes = [vp, u64, vp]
    L.bdr_agent_param_count_of.argtypes = [vp, i32, C.POINTER(u64)]
    L.bdr_dqn_update_on_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, C.POINTER(DqnRecordC)]
    L.bdr_agent_qvalues.argtypes = [vp, u64, vp, vp, vp]
    L.bdr_agent_sample_device.argtypes = [vp, u64, vp, u64, vp, vp]
    L.bdr_agent_qvalues_device.argtypes = [vp, u64, vp, u64, vp, vp]
    L.bdr_sac_sample_device.argtypes = [vp, u64, vp, u64, vp]
    L.bdr_agent_sync.argtypes = [vp]
    L.bdr_agent_n_opts.argtypes = [vp, C.POINTER(u64)]
    L.bdr_agent_param_count.argtypes = [vp, C.POINTER(u64)]
    L.bdr_agent_get_params.argtypes = [vp, i32, vp, u64]
    L.bdr_agent_set_params.argtypes = [vp, i32, vp, u64]
    L.bdr_agent_arena_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u64)]
    L.bdr_agent_arena_release.argtypes = [vp, i32]
    L.bdr_agent_save_params.argtypes = [vp, C.c_char_p]
    L.bdr_agent_load_params.argtypes = [vp, C.c_char_p]
    L.bdr_dqn_probe.argtypes = [vp, i32, vp, u64]
    L.bdr_agent_profile_enable.argtypes = [vp, i32]
    L.bdr_agent_profile_read.argtypes = [vp, vp, u64, vp, C.POINTER(u64)]
    L.bdr_iqn_config_default.argtypes = [C.POINTER(IqnConfigC)]
    L.bdr_iqn_create.argtypes = [C.POINTER(IqnConfigC), C.POINTER(vp)]
    L.bdr_iqn_update_on_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp]
    L.bdr_iqn_forward.argtypes = [vp, i32, u64, vp, vp, i32, vp]
    L.bdr_iqn_qvalues.argtypes = [vp, u64, vp, vp, vp]
    L.bdr_sac_config_default.argtypes = [C.POINTER(SacConfigC)]
    L.bdr_sac_create.argtypes = [C.POINTER(SacConfigC), C.POINTER(vp)]
    L.bdr_sac_update_on_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.bdr_sac_sample.argtypes = [vp, u64, vp, vp]
    L.bdr_sac_probe.argtypes = [vp, i32, vp, u64]
    L.bdr_comm_get_unique_id.argtypes = [vp]
    L.bdr_comm_init_rank.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.bdr_comm_destroy.argtypes = [vp]
    L.bdr_comm_agree.argtypes = [vp, i32, C.POINTER(i32)]
    L.bdr_agent_allreduce_params.argtypes = [vp, vp, i32]
    L.bdr_agent_broadcast_params.argtypes = [vp, vp, i32, i32]
    L.bdr_agent_set_grad_comm.argtypes = [vp, vp]
    L.bdr_dqn_grads_on_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, C.POINTER(DqnRecordC)]
    L.bdr_agent_apply_grads.argtypes = [vp]
    _lib = L
    return L


def check(status: int) -> None:
    if status != 0:
        raise BdrError(status, lib().bdr_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int32(0)
    lib().bdr_device_count(C.byref(n))
    return n.value
