"""Host-side mirror of border-tch-agent's Sac agent over the C ABI.

  SacConfig     border-tch-agent/src/sac/config.rs (defaults :85-105)
  EntCoefMode   border-tch-agent/src/sac/ent_coef.rs:14-25  (Fix(alpha) | Auto(target_entropy, lr))
  Sac           border-tch-agent/src/sac/base.rs (Agent, Policy::sample, SyncModel ships `pi` only)
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .dqn import OptimizerConfig
from .replay import SimpleReplayBuffer


@dataclass
class SacConfig:
    obs_dim: int = 0
    act_dim: int = 0
    pi_units: Tuple[int, ...] = (64, 64)        # ActorConfig.pi_config: MlpConfig(in, units, out)
    q_units: Tuple[int, ...] = (64, 64)         # CriticConfig.q_config
    lr_actor: float = 3e-4
    lr_critic: float = 3e-4
    opt_actor: Optional[OptimizerConfig] = None   # ActorConfig.opt_config (sac/actor/config.rs:15); None = OptimizerConfig.Adam(lr_actor)
    opt_critic: Optional[OptimizerConfig] = None  # CriticConfig.opt_config; None = OptimizerConfig.Adam(lr_critic)
    gamma: float = 0.99
    tau: float = 0.005
    ent_coef_mode: tuple = ("Fix", 1.0)          # or ("Auto", target_entropy, lr)
    epsilon: float = 1e-4
    min_lstd: float = -20.0
    max_lstd: float = 2.0
    n_updates_per_opt: int = 1
    batch_size: int = 1
    train: bool = False
    critic_loss: str = "Mse"
    reward_scale: float = 1.0
    n_critics: int = 1
    seed: int = 0
    device: Optional[int] = None

    def to_c(self) -> _lib.SacConfigC:
        c = _lib.SacConfigC()
        _lib.lib().bdr_sac_config_default(C.byref(c))
        c.obs_dim, c.act_dim = self.obs_dim, self.act_dim
        c.n_pi_units, c.n_q_units = len(self.pi_units), len(self.q_units)
        for i, u in enumerate(self.pi_units):
            c.pi_units[i] = u
        for i, u in enumerate(self.q_units):
            c.q_units[i] = u
        c.lr_actor, c.lr_critic, c.gamma, c.tau = self.lr_actor, self.lr_critic, self.gamma, self.tau
        if self.ent_coef_mode[0] == "Auto":
            c.ent_coef_auto, c.target_entropy, c.ent_coef_lr = 1, self.ent_coef_mode[1], self.ent_coef_mode[2]
        else:
            c.ent_coef_auto, c.ent_coef_alpha = 0, self.ent_coef_mode[1]
        c.epsilon, c.min_lstd, c.max_lstd = self.epsilon, self.min_lstd, self.max_lstd
        c.n_updates_per_opt, c.batch_size, c.train = self.n_updates_per_opt, self.batch_size, int(self.train)
        c.critic_loss = {"Mse": 0, "SmoothL1": 1}[self.critic_loss]
        c.reward_scale, c.n_critics, c.seed = self.reward_scale, self.n_critics, self.seed
        c.device = -1 if self.device is None else self.device
        if self.opt_actor is not None:
            c.lr_actor = self.opt_actor.lr
            c.opt_actor.fill(self.opt_actor)
        if self.opt_critic is not None:
            c.lr_critic = self.opt_critic.lr
            c.opt_critic.fill(self.opt_critic)
        return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Sac:
    def __init__(self, config: SacConfig):
        self.config = config
        h = C.c_void_p()
        c = config.to_c()
        _lib.check(_lib.lib().bdr_sac_create(C.byref(c), C.byref(h)))
        self._h = h

    @classmethod
    def build(cls, config: SacConfig) -> "Sac":
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

    # model ids (bdr_agent_get_params `which`)
    def which(self, name: str, role: str = "param") -> int:
        nc = self.config.n_critics
        base = {"pi": 0, "log_alpha": 1 + 2 * nc}
        if name.startswith("qnet_tgt_"):
            i = 1 + nc + int(name[len("qnet_tgt_"):])
        elif name.startswith("qnet_"):
            i = 1 + int(name[len("qnet_"):])
        else:
            i = base[name]
        return i + {"param": 0, "grad": 100, "exp_avg": 200, "exp_avg_sq": 300, "max_exp_avg_sq": 400}[role]

    WHICH = {"qnet": 0, "pi": 0}   # ParamExchange / ModelMailbox: SyncModel ships `pi` (sac/base.rs:377-386) == model 0

    def arena_device_ptr(self, which="pi"):
        """(device pointer, float count) of a flat parameter arena in the kernels' internal layout."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_arena_device_ptr(self._h, self.WHICH[which], C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

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

    def update_on_batch(self, obs, act, next_obs, reward, is_terminated, z_actor, z_next) -> dict:
        f = lambda x: np.ascontiguousarray(x, dtype=np.float32)
        obs, act, next_obs, reward, z_actor, z_next = map(f, (obs, act, next_obs, reward, z_actor, z_next))
        term = np.ascontiguousarray(is_terminated, dtype=np.int8)
        rec = np.zeros(3, np.float32)
        _lib.check(_lib.lib().bdr_sac_update_on_batch(self._h, len(reward), _p(obs), _p(act), _p(next_obs), _p(reward), _p(term),
                                                      _p(z_actor), _p(z_next), _p(rec)))
        return dict(loss_critic=float(rec[0]), loss_actor=float(rec[1]), ent_coef=float(rec[2]))

    PROBES = {"q_pred": 0, "q_next": 1, "qvals_min": 2, "next_log_p": 3, "tgt": 4, "q_pi": 5, "log_p": 6, "next_act": 7}

    def probe(self, what: str, batch: int) -> np.ndarray:
        """Intermediates of the last update (bdr_sac_probe): q_pred / q_next / q_pi [n_critics, B], qvals_min / next_log_p / tgt /
        log_p [B], next_act [B, act_dim]."""
        nc, ad = self.config.n_critics, self.config.act_dim
        shape = {"q_pred": (nc, batch), "q_next": (nc, batch), "q_pi": (nc, batch), "next_act": (batch, ad)}.get(what, (batch,))
        out = np.empty(shape, np.float32)
        _lib.check(_lib.lib().bdr_sac_probe(self._h, self.PROBES[what], _p(out), out.size))
        return out

    def sample(self, obs) -> np.ndarray:
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        out = np.empty((obs.shape[0], self.config.act_dim), np.float32)
        _lib.check(_lib.lib().bdr_sac_sample(self._h, obs.shape[0], _p(obs), _p(out)))
        return out

    def sample_device(self, obs_dev: int, n: int, row_stride: int) -> np.ndarray:
        """`sample` for observation rows already in HBM (`bdr_sac_sample_device`): row i at obs_dev + i * row_stride bytes."""
        out = np.empty((n, self.config.act_dim), np.float32)
        _lib.check(_lib.lib().bdr_sac_sample_device(self._h, n, C.c_void_p(obs_dev), row_stride, _p(out)))
        return out

    def sync(self):
        _lib.check(_lib.lib().bdr_agent_sync(self._h))

    @property
    def n_opts(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_n_opts(self._h, C.byref(n)))
        return n.value

    def param_count(self, name="pi") -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_agent_param_count_of(self._h, self.which(name), C.byref(n)))
        return n.value

    def get_params(self, name="pi", role="param") -> np.ndarray:
        out = np.empty(self.param_count(name), np.float32)
        _lib.check(_lib.lib().bdr_agent_get_params(self._h, self.which(name, role), _p(out), out.size))
        return out

    def set_params(self, params, name="pi", role="param") -> None:
        p = np.ascontiguousarray(params, dtype=np.float32).reshape(-1)
        _lib.check(_lib.lib().bdr_agent_set_params(self._h, self.which(name, role), _p(p), p.size))

    def model_info(self):
        """SyncModel::model_info (sac/base.rs:377-383): only the policy network."""
        return self.n_opts, self.get_params("pi")

    def sync_model(self, model_info) -> None:
        self.set_params(model_info, "pi")

    def set_checkpoint_format(self, fmt: str) -> None:
        """"tch" (default): `<stem>.pt.tch` libtorch archives, the reference's files; "safetensors": `<stem>.safetensors`."""
        from .checkpoint import FORMATS
        _lib.check(_lib.lib().bdr_agent_set_checkpoint_format(self._h, FORMATS[fmt]))
        self._ckpt_ext = {"tch": ".pt.tch", "safetensors": ".safetensors"}[fmt]

    def save_params(self, path: str):
        os.makedirs(path, exist_ok=True)
        _lib.check(_lib.lib().bdr_agent_save_params(self._h, path.encode()))
        nc = self.config.n_critics
        ext = getattr(self, "_ckpt_ext", ".pt.tch")
        stems = [s for i in range(nc) for s in (f"qnet_{i}", f"qnet_tgt_{i}")] + ["pi", "ent_coef"]   # sac/base.rs:313-334 order
        return [os.path.join(path, stem + ext) for stem in stems]

    def load_params(self, path: str):
        _lib.check(_lib.lib().bdr_agent_load_params(self._h, path.encode()))
