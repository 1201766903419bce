"""Host-side mirror of border-tch-agent's Iqn agent over the C ABI.

  IqnConfig       border-tch-agent/src/iqn/config.rs (defaults :50-67)
  IqnModelConfig  border-tch-agent/src/iqn/model/config.rs (feature_dim, embed_dim, opt_config)
  IqnSample       border-tch-agent/src/iqn/model/base.rs:327-387
  Iqn             border-tch-agent/src/iqn/base.rs
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .dqn import AtariCnnConfig, MlpConfig, OptimizerConfig
from .replay import SimpleReplayBuffer

IQN_SAMPLE = {"Const10": 0, "Const32": 1, "Uniform10": 2, "Uniform8": 3, "Uniform32": 4, "Uniform64": 5, "Median": 6}


@dataclass
class IqnConfig:
    f_config: object = field(default_factory=lambda: AtariCnnConfig(n_stack=4, out_dim=0, skip_linear=True))
    feature_dim: int = 3136
    embed_dim: int = 64
    m_units: Tuple[int, ...] = (512,)            # merge net M = Mlp(feature_dim, units, n_actions)
    n_actions: int = 0
    lr: float = 1e-4
    opt_config: Optional[OptimizerConfig] = None   # IqnModelConfig.opt_config (iqn/model/config.rs:50); None = OptimizerConfig.Adam(lr)
    soft_update_interval: int = 1
    n_updates_per_opt: int = 1
    batch_size: int = 1
    discount_factor: float = 0.99
    tau: float = 0.005
    sample_percents_pred: str = "Uniform8"
    sample_percents_tgt: str = "Uniform8"
    sample_percents_act: str = "Const32"
    train: bool = False
    device: Optional[int] = None
    seed: int = 0
    arithmetic: str = "bf16x3_6"                 # BDR_ARITH_* (not a reference field): "bf16x3_6" | "f32_exact"

    def to_c(self) -> _lib.IqnConfigC:
        c = _lib.IqnConfigC()
        _lib.lib().bdr_iqn_config_default(C.byref(c))
        q = self.f_config
        if isinstance(q, AtariCnnConfig):
            c.psi.kind, c.psi.n_stack = 0, q.n_stack
        else:
            c.psi.kind, c.psi.in_dim, c.psi.n_units, c.psi.out_dim = 1, q.in_dim, len(q.units), q.out_dim
            for i, u in enumerate(q.units):
                c.psi.units[i] = u
            c.psi.activation_out = int(q.activation_out)
        c.feature_dim, c.embed_dim, c.n_f_units, c.n_actions, c.lr = self.feature_dim, self.embed_dim, len(self.m_units), self.n_actions, self.lr
        for i, u in enumerate(self.m_units):
            c.f_units[i] = u
        c.soft_update_interval, c.n_updates_per_opt, c.batch_size = self.soft_update_interval, self.n_updates_per_opt, self.batch_size
        c.discount_factor, c.tau = self.discount_factor, self.tau
        c.sample_percents_pred, c.sample_percents_tgt, c.sample_percents_act = (IQN_SAMPLE[self.sample_percents_pred],
                                                                               IQN_SAMPLE[self.sample_percents_tgt],
                                                                               IQN_SAMPLE[self.sample_percents_act])
        c.train, c.device, c.seed = int(self.train), -1 if self.device is None else self.device, self.seed
        if self.opt_config is not None:
            c.lr = self.opt_config.lr
            c.opt.fill(self.opt_config)
        c.arithmetic = _lib.ARITHMETIC[self.arithmetic]
        return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Iqn:
    WHICH = {"qnet": 0, "iqn": 0, "iqn_tgt": 1, "exp_avg": 2, "exp_avg_sq": 3, "grad": 4, "max_exp_avg_sq": 5}

    def arena_device_ptr(self, which="iqn"):
        """(device pointer, float count) of a flat parameter arena in the kernels' internal layout."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_arena_device_ptr(self._h, self.WHICH[which], C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def __init__(self, config: IqnConfig):
        self.config = config
        h = C.c_void_p()
        c = config.to_c()
        _lib.check(_lib.lib().bdr_iqn_create(C.byref(c), C.byref(h)))
        self._h = h

    @classmethod
    def build(cls, config: IqnConfig) -> "Iqn":
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

    def train(self):
        _lib.check(_lib.lib().bdr_agent_set_train(self._h, 1))

    def eval(self):
        _lib.check(_lib.lib().bdr_agent_set_train(self._h, 0))

    def opt(self, buffer: SimpleReplayBuffer) -> None:
        _lib.check(_lib.lib().bdr_agent_opt(self._h, buffer.handle))

    def opt_with_record(self, buffer: SimpleReplayBuffer) -> dict:
        from .dqn import opt_with_named_record
        return opt_with_named_record(self._h, buffer)

    def profile_enable(self, on: bool = True):
        _lib.check(_lib.lib().bdr_agent_profile_enable(self._h, int(on)))

    def draw_noise(self, n: int) -> np.ndarray:
        """n draws of the agent's device noise stream (test helper, bdr_agent_draw_noise)."""
        from .dqn import draw_noise
        return draw_noise(self._h, n)

    def update_on_batch(self, obs, act, next_obs, reward, is_terminated, tau_pred, tau_tgt) -> dict:
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        n = len(reward)
        obs, next_obs = np.ascontiguousarray(obs), np.ascontiguousarray(next_obs)
        act = np.ascontiguousarray(act, dtype=np.int64).reshape(n)
        term = np.ascontiguousarray(is_terminated, dtype=np.int8)
        tp, tt = np.ascontiguousarray(tau_pred, dtype=np.float32), np.ascontiguousarray(tau_tgt, dtype=np.float32)
        loss = np.zeros(1, np.float32)
        _lib.check(_lib.lib().bdr_iqn_update_on_batch(self._h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term),
                                                      _p(tp), tp.shape[1], _p(tt), tt.shape[1], _p(loss)))
        return dict(loss_critic=float(loss[0]))

    def forward(self, obs, tau, which="iqn") -> np.ndarray:
        obs, tau = np.ascontiguousarray(obs), np.ascontiguousarray(tau, dtype=np.float32)
        n, nt = tau.shape
        z = np.empty((n, nt, self.config.n_actions), np.float32)
        _lib.check(_lib.lib().bdr_iqn_forward(self._h, self.WHICH[which], n, _p(obs), _p(tau), nt, _p(z)))
        return z

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
        q = np.empty((n, self.config.n_actions), np.float32)
        _lib.check(_lib.lib().bdr_agent_qvalues_device(self._h, n, C.c_void_p(obs_dev), row_stride, _p(q), None))
        return q

    def qvalues(self, obs) -> np.ndarray:
        obs = np.ascontiguousarray(obs)
        q = np.empty((obs.shape[0], self.config.n_actions), np.float32)
        _lib.check(_lib.lib().bdr_iqn_qvalues(self._h, obs.shape[0], _p(obs), _p(q), None))
        return q

    def set_explorer(self, explorer, seed: int = 0) -> None:
        """IqnConfig::explorer (iqn/config.rs:35; default Softmax, :62)."""
        _lib.check(_lib.lib().bdr_agent_set_explorer(self._h, C.byref(explorer.to_c(seed))))

    def sample(self, obs, return_info: bool = False):
        """Policy::sample (iqn/base.rs:204-228): quantile-averaged action values + exploration (train) / argmax (eval)."""
        obs = np.ascontiguousarray(obs)
        n = obs.shape[0]
        a = np.empty(n, np.int64)
        info = _lib.SampleInfoC()
        _lib.check(_lib.lib().bdr_agent_sample(self._h, n, _p(obs), _p(a), C.byref(info)))
        if return_info:
            return a, {"eps": info.eps, "is_random": bool(info.is_random), "n_samples_act": info.n_samples_act,
                       "n_samples_best_act": info.n_samples_best_act}
        return a

    def sync(self):
        _lib.check(_lib.lib().bdr_agent_sync(self._h))

    @property
    def n_opts(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_n_opts(self._h, C.byref(n)))
        return n.value

    def param_count(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_param_count(self._h, C.byref(n)))
        return n.value

    def get_params(self, which="iqn") -> np.ndarray:
        out = np.empty(self.param_count(), np.float32)
        _lib.check(_lib.lib().bdr_agent_get_params(self._h, self.WHICH[which], _p(out), out.size))
        return out

    def set_params(self, params, which="iqn") -> None:
        p = np.ascontiguousarray(params, dtype=np.float32)
        _lib.check(_lib.lib().bdr_agent_set_params(self._h, self.WHICH[which], _p(p), p.size))

    def set_checkpoint_format(self, fmt: str) -> None:
        """"tch" (default): `<stem>.pt.tch` libtorch archives, the reference's files; "safetensors": `<stem>.safetensors`."""
        from .checkpoint import FORMATS
        _lib.check(_lib.lib().bdr_agent_set_checkpoint_format(self._h, FORMATS[fmt]))
        self._ckpt_ext = {"tch": ".pt.tch", "safetensors": ".safetensors"}[fmt]

    def save_params(self, path: str):
        os.makedirs(path, exist_ok=True)
        _lib.check(_lib.lib().bdr_agent_save_params(self._h, path.encode()))
        ext = getattr(self, "_ckpt_ext", ".pt.tch")
        return [os.path.join(path, stem + ext) for stem in ["iqn", "iqn_tgt"]]

    def load_params(self, path: str):
        _lib.check(_lib.lib().bdr_agent_load_params(self._h, path.encode()))
