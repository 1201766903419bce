"""Host-side mirror of border-tch-agent's Dqn agent over the C ABI.

Same names / argument meaning as the reference:
  DqnConfig        border-tch-agent/src/dqn/config.rs:26-48 (defaults :82-102)
  DqnModelConfig   border-tch-agent/src/dqn/model/config.rs (q_config + opt_config)
  AtariCnnConfig   border-tch-agent/src/cnn/config.rs:13-18
  OptimizerConfig  border-tch-agent/src/opt.rs:13-28
  Dqn              border-tch-agent/src/dqn/base.rs  (Agent: train/eval/is_train/opt/
                   opt_with_record/save_params/load_params; Policy::sample; SyncModel)
All arithmetic runs in the HIP library; there is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .replay import SimpleReplayBuffer


@dataclass
class AtariCnnConfig:
    n_stack: int = 4
    out_dim: int = 0
    skip_linear: bool = False


@dataclass
class MlpConfig:
    in_dim: int = 0
    units: Tuple[int, ...] = ()
    out_dim: int = 0
    activation_out: bool = False


@dataclass
class OptimizerConfig:
    """opt.rs:13-28: Adam{lr} or AdamW{lr,beta1,beta2,wd,eps,amsgrad}."""
    kind: str = "Adam"
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    wd: float = 0.0
    eps: float = 1e-8
    amsgrad: bool = False

    @classmethod
    def Adam(cls, lr: float) -> "OptimizerConfig":
        return cls("Adam", lr)

    @classmethod
    def AdamW(cls, lr: float, beta1: float = 0.9, beta2: float = 0.999, wd: float = 0.01, eps: float = 1e-8,
              amsgrad: bool = False) -> "OptimizerConfig":
        return cls("AdamW", lr, beta1, beta2, wd, eps, amsgrad)


@dataclass
class Softmax:
    """dqn/explorer.rs:17-32 (the DqnConfig default, dqn/config.rs:93)."""

    def to_c(self, seed: int = 0) -> "_lib.ExplorerConfigC":
        e = _lib.ExplorerConfigC()
        _lib.lib().bdr_explorer_config_default(C.byref(e), 0)
        e.seed = seed
        return e


@dataclass
class EpsilonGreedy:
    """dqn/explorer.rs:34-120: eps decays linearly over `final_step` action() calls."""
    n_opts: int = 0
    eps_start: float = 1.0
    eps_final: float = 0.02
    final_step: int = 100_000

    @classmethod
    def with_final_step(cls, final_step: int) -> "EpsilonGreedy":
        return cls(final_step=final_step)

    def to_c(self, seed: int = 0) -> "_lib.ExplorerConfigC":
        e = _lib.ExplorerConfigC()
        _lib.lib().bdr_explorer_config_default(C.byref(e), 1)
        e.eps_start, e.eps_final, e.final_step, e.n_calls, e.seed = self.eps_start, self.eps_final, self.final_step, self.n_opts, seed
        return e


@dataclass
class DqnModelConfig:
    q_config: Optional[object] = None
    opt_config: OptimizerConfig = field(default_factory=lambda: OptimizerConfig.Adam(0.0))


@dataclass
class DqnConfig:
    """dqn/config.rs:26-48 with the defaults of :82-102."""
    model_config: DqnModelConfig = field(default_factory=DqnModelConfig)
    soft_update_interval: int = 1
    n_updates_per_opt: int = 1
    batch_size: int = 1
    discount_factor: float = 0.99
    tau: float = 0.005
    train: bool = False
    clip_reward: Optional[float] = None     # never used by the reference either (dqn/base.rs:42,276)
    double_dqn: bool = False
    clip_td_err: Optional[Tuple[float, float]] = None
    device: Optional[int] = None            # None -> "No device is given for DQN agent" (dqn/base.rs:256)
    critic_loss: str = "Mse"                # util.rs:17-23
    record_verbose_level: int = 0
    param_seed: int = 0
    arithmetic: str = "bf16x3_6"            # BDR_ARITH_* (include/border_amd.h; not a reference field): "bf16x3_6" | "f32_exact"

    def to_c(self) -> _lib.DqnConfigC:
        c = _lib.DqnConfigC()
        _lib.lib().bdr_dqn_config_default(C.byref(c))
        q = self.model_config.q_config
        if isinstance(q, AtariCnnConfig):
            if q.skip_linear:
                raise NotImplementedError("skip_linear AtariCnn is only used by IQN (not built yet)")
            c.net.kind, c.net.n_stack, c.net.out_dim = 0, q.n_stack, q.out_dim
        elif isinstance(q, MlpConfig):
            c.net.kind, c.net.in_dim, c.net.n_units, c.net.out_dim = 1, q.in_dim, len(q.units), q.out_dim
            for i, u in enumerate(q.units):
                c.net.units[i] = u
            c.net.activation_out = int(q.activation_out)
        else:
            raise ValueError("model_config.q_config must be an AtariCnnConfig or MlpConfig")
        o = self.model_config.opt_config
        c.opt_kind = {"Adam": 0, "AdamW": 1}[o.kind]
        c.lr, c.beta1, c.beta2, c.weight_decay, c.eps = o.lr, o.beta1, o.beta2, o.wd, o.eps
        c.amsgrad = 1 if (o.kind == "AdamW" and o.amsgrad) else 0
        c.soft_update_interval, c.n_updates_per_opt, c.batch_size = (self.soft_update_interval,
                                                                    self.n_updates_per_opt, self.batch_size)
        c.discount_factor, c.tau, c.train = self.discount_factor, self.tau, int(self.train)
        c.double_dqn = int(self.double_dqn)
        c.critic_loss = {"Mse": 0, "SmoothL1": 1}[self.critic_loss]
        if self.clip_td_err is not None:
            c.has_clip_td_err, (c.clip_td_err_min, c.clip_td_err_max) = 1, self.clip_td_err
        c.record_verbose_level = self.record_verbose_level
        c.device = -1 if self.device is None else self.device
        c.param_seed = self.param_seed
        c.arithmetic = _lib.ARITHMETIC[self.arithmetic]
        return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def record_keys(handle) -> list:
    names = C.create_string_buffer(8192)
    n = C.c_int32()
    _lib.check(_lib.lib().bdr_agent_record_keys(handle, names, 8192, C.byref(n)))
    return names.value.decode().split("\n")[:n.value]


def opt_with_named_record(handle, buffer) -> dict:
    """bdr_agent_opt_with_scalars + bdr_agent_record_keys: the reference's Record of one opt_with_record call."""
    out = np.zeros(128, np.float32)
    n = C.c_int32()
    _lib.check(_lib.lib().bdr_agent_opt_with_scalars(handle, buffer.handle, _p(out), 128, C.byref(n)))
    return {k: float(v) for k, v in zip(record_keys(handle), out[:n.value])}


def draw_noise(handle, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    _lib.check(_lib.lib().bdr_agent_draw_noise(handle, n, _p(out)))
    return out


class Dqn:
    """Dqn<E, Q, R> (dqn/base.rs:22-48)."""

    def __init__(self, config: DqnConfig):
        self.config = config
        h = C.c_void_p()
        c = config.to_c()
        _lib.check(_lib.lib().bdr_dqn_create(C.byref(c), C.byref(h)))
        self._h = h
        self.n_actions = config.model_config.q_config.out_dim

    @classmethod
    def build(cls, config: DqnConfig) -> "Dqn":   # Configurable::build (dqn/base.rs:252-286)
        return cls(config)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().bdr_agent_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # Agent ----------------------------------------------------------------------------------
    def train(self):
        _lib.check(_lib.lib().bdr_agent_set_train(self._h, 1))

    def eval(self):
        _lib.check(_lib.lib().bdr_agent_set_train(self._h, 0))

    def is_train(self) -> bool:
        v = C.c_int32()
        _lib.check(_lib.lib().bdr_agent_is_train(self._h, C.byref(v)))
        return bool(v.value)

    def opt(self, buffer: SimpleReplayBuffer) -> None:
        """Agent::opt -- enqueue one optimisation step; does not synchronise."""
        _lib.check(_lib.lib().bdr_agent_opt(self._h, buffer.handle))

    def opt_with_record(self, buffer: SimpleReplayBuffer) -> dict:
        """Agent::opt_with_record (dqn/base.rs:316-342): the Record as a dict - "loss"; with record_verbose_level >= 2 also
        pred_mean / reward_mean / tgt_mean / tgt_minus_pred_mean, `<var>_mean` / `<var>_std` of every qnet variable
        (param_stats, util.rs:64-80) and ratio_best_act (which resets the sample counters)."""
        return opt_with_named_record(self._h, buffer)

    @staticmethod
    def _record(r) -> dict:
        rec = {"loss": r.loss}
        if r.has_verbose:
            rec.update(pred_mean=r.pred_mean, reward_mean=r.reward_mean, tgt_mean=r.tgt_mean,
                       tgt_minus_pred_mean=r.tgt_minus_pred_mean)
        return rec

    def update_on_batch(self, obs, act, next_obs, reward, is_terminated, weight=None) -> dict:
        """One Dqn::opt_ on a caller-supplied minibatch (fixed-minibatch parity tests).  With `weight` the
        importance-weighted branch of update_critic runs (dqn/base.rs:123-145) and the record carries `td_errs`."""
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        n = len(reward)
        obs, next_obs = np.ascontiguousarray(obs), np.ascontiguousarray(next_obs)
        act = np.ascontiguousarray(act, dtype=np.int64).reshape(n)
        term = np.ascontiguousarray(is_terminated, dtype=np.int8)
        r = _lib.DqnRecordC()
        if weight is None:
            _lib.check(_lib.lib().bdr_dqn_update_on_batch(self._h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term),
                                                          C.byref(r)))
            return self._record(r)
        w = np.ascontiguousarray(weight, dtype=np.float32)
        td = np.empty(n, np.float32)
        _lib.check(_lib.lib().bdr_dqn_update_on_batch_weighted(self._h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term),
                                                               _p(w), _p(td), C.byref(r)))
        rec = self._record(r)
        rec["td_errs"] = td
        return rec

    def grads_on_batch(self, obs, act, next_obs, reward, is_terminated) -> dict:
        """update_critic up to loss.backward() on a host minibatch: gradients -> get_params("grad"); nothing else changes."""
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        n = len(reward)
        obs, next_obs = np.ascontiguousarray(obs), np.ascontiguousarray(next_obs)
        act = np.ascontiguousarray(act, dtype=np.int64).reshape(n)
        term = np.ascontiguousarray(is_terminated, dtype=np.int8)
        r = _lib.DqnRecordC()
        _lib.check(_lib.lib().bdr_dqn_grads_on_batch(self._h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term), C.byref(r)))
        return self._record(r)

    def apply_grads(self) -> None:
        """The optimizer step on the gradient arena + opt_'s bookkeeping (soft update counter, n_opts)."""
        _lib.check(_lib.lib().bdr_agent_apply_grads(self._h))

    def sync(self):
        _lib.check(_lib.lib().bdr_agent_sync(self._h))

    @property
    def n_opts(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_n_opts(self._h, C.byref(n)))
        return n.value

    def set_checkpoint_format(self, fmt: str) -> None:
        """"tch" (default): `<stem>.pt.tch` libtorch archives, the reference's files; "safetensors": `<stem>.safetensors`."""
        from .checkpoint import FORMATS
        _lib.check(_lib.lib().bdr_agent_set_checkpoint_format(self._h, FORMATS[fmt]))
        self._ckpt_ext = {"tch": ".pt.tch", "safetensors": ".safetensors"}[fmt]

    def save_params(self, path: str):
        os.makedirs(path, exist_ok=True)
        _lib.check(_lib.lib().bdr_agent_save_params(self._h, path.encode()))
        ext = getattr(self, "_ckpt_ext", ".pt.tch")
        return [os.path.join(path, stem + ext) for stem in ["qnet", "qnet_tgt"]]

    def load_params(self, path: str):
        _lib.check(_lib.lib().bdr_agent_load_params(self._h, path.encode()))

    # Policy ----------------------------------------------------------------------------------------
    def set_explorer(self, explorer: "Softmax | EpsilonGreedy", seed: int = 0) -> None:
        """DqnConfig::explorer (dqn/config.rs:45); `seed` seeds the library's exploration stream."""
        _lib.check(_lib.lib().bdr_agent_set_explorer(self._h, C.byref(explorer.to_c(seed))))

    def explorer_state(self) -> dict:
        e = _lib.ExplorerConfigC()
        _lib.check(_lib.lib().bdr_agent_get_explorer(self._h, C.byref(e)))
        return {"kind": "softmax" if e.kind == 0 else "eps_greedy", "eps_start": e.eps_start, "eps_final": e.eps_final,
                "final_step": e.final_step, "n_opts": e.n_calls}

    def sample(self, obs, return_info: bool = False):
        """Policy::sample (dqn/base.rs:211-242): forward on the device + the configured exploration.
        obs: [n_procs, ...] rows as stored in the replay buffer; returns int64 actions [n_procs]."""
        obs = np.ascontiguousarray(obs)
        n = obs.shape[0]
        a = np.empty(n, np.int64)
        info = _lib.SampleInfoC()
        _lib.check(_lib.lib().bdr_agent_sample(self._h, n, _p(obs), _p(a), C.byref(info)))
        if return_info:
            return a, {"eps": info.eps, "is_random": bool(info.is_random), "n_samples_act": info.n_samples_act,
                       "n_samples_best_act": info.n_samples_best_act}
        return a

    def sample_device(self, obs_dev: int, n: int, row_stride: int, return_info: bool = False):
        """Policy::sample for observation rows already in HBM (`bdr_agent_sample_device`): row i at obs_dev + i * row_stride bytes."""
        a = np.empty(n, np.int64)
        info = _lib.SampleInfoC()
        _lib.check(_lib.lib().bdr_agent_sample_device(self._h, n, C.c_void_p(obs_dev), row_stride, _p(a), C.byref(info)))
        if return_info:
            return a, {"eps": info.eps, "is_random": bool(info.is_random), "n_samples_act": info.n_samples_act,
                       "n_samples_best_act": info.n_samples_best_act}
        return a

    def qvalues_device(self, obs_dev: int, n: int, row_stride: int) -> np.ndarray:
        q = np.empty((n, self.n_actions), np.float32)
        _lib.check(_lib.lib().bdr_agent_qvalues_device(self._h, n, C.c_void_p(obs_dev), row_stride, _p(q), None))
        return q

    def qvalues(self, obs) -> np.ndarray:
        obs = np.ascontiguousarray(obs)
        n = obs.shape[0]
        q = np.empty((n, self.n_actions), np.float32)
        _lib.check(_lib.lib().bdr_agent_qvalues(self._h, n, _p(obs), _p(q), None))
        return q

    def sample_greedy(self, obs) -> np.ndarray:
        obs = np.ascontiguousarray(obs)
        n = obs.shape[0]
        a = np.empty(n, np.int64)
        _lib.check(_lib.lib().bdr_agent_qvalues(self._h, n, _p(obs), None, _p(a)))
        return a

    # SyncModel / parameter access --------------------------------------------------------------
    WHICH = {"qnet": 0, "qnet_tgt": 1, "exp_avg": 2, "exp_avg_sq": 3, "grad": 4, "max_exp_avg_sq": 5}

    def arena_device_ptr(self, which="qnet"):
        """(device pointer, float count) of the flat parameter arena in the kernels' internal layout."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_arena_device_ptr(self._h, self.WHICH[which], C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def arena_release(self, which="qnet") -> None:
        """hands an arena back after arena_device_ptr: the library's derived copies (bf16 weight planes) are trusted again"""
        _lib.check(_lib.lib().bdr_agent_arena_release(self._h, self.WHICH[which]))

    def param_count(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_param_count(self._h, C.byref(n)))
        return n.value

    def get_params(self, which="qnet") -> np.ndarray:
        out = np.empty(self.param_count(), np.float32)
        _lib.check(_lib.lib().bdr_agent_get_params(self._h, self.WHICH[which], _p(out), out.size))
        return out

    def set_params(self, params: np.ndarray, which="qnet") -> None:
        p = np.ascontiguousarray(params, dtype=np.float32)
        _lib.check(_lib.lib().bdr_agent_set_params(self._h, self.WHICH[which], _p(p), p.size))

    def model_info(self):
        """SyncModel::model_info (dqn/base.rs:377-389): (n_opts, qnet parameters)."""
        return self.n_opts, self.get_params("qnet")

    def sync_model(self, model_info) -> None:
        """SyncModel::sync_model (dqn/base.rs:391-402)."""
        self.set_params(model_info, "qnet")

    # probes / profiling ----------------------------------------------------------------------
    def probe(self, what: str, n: int) -> np.ndarray:
        idx = {"q_pred_all": 0, "q_next_all": 1, "pred": 2, "tgt": 3, "loss": 4, "act_conv1": 5, "act_conv2": 6, "act_conv3": 7}[what]
        out = np.empty(n, np.float32)
        _lib.check(_lib.lib().bdr_dqn_probe(self._h, idx, _p(out), n))
        return out

    def profile_enable(self, on: bool = True):
        _lib.check(_lib.lib().bdr_agent_profile_enable(self._h, int(on)))

    def profile_read(self) -> dict:
        cnt = C.c_uint64(256)
        names = C.create_string_buffer(8192)
        ms = np.zeros(256, np.float32)
        _lib.check(_lib.lib().bdr_agent_profile_read(self._h, names, 8192, _p(ms), C.byref(cnt)))
        labels = names.value.decode().split("\n")[:cnt.value]
        return {l: float(ms[i]) for i, l in enumerate(labels)}
