"""ctypes binding of the CPU oracle (oracle/border_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of border_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from border_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libborder_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "border_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class RngState(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("word_pos", C.c_uint64)]


class NetCfg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("n_stack", C.c_int32),
        ("in_dim", C.c_int32),
        ("n_units", C.c_int32),
        ("units", C.c_int32 * 8),
        ("out_dim", C.c_int32),
        ("activation_out", C.c_int32),
    ]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("step", C.c_int64)]


class DqnCfg(C.Structure):
    _fields_ = [("discount_factor", C.c_double), ("double_dqn", C.c_int32),
                ("critic_loss", C.c_int32), ("has_clip_td_err", C.c_int32),
                ("clip_min", C.c_double), ("clip_max", C.c_double)]


class DqnProbe(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("q_pred_all", "q_next_all", "pred", "tgt", "grads", "td_abs")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_rng_next_u32.restype = C.c_uint32
        L.orc_rng_next_u64.restype = C.c_uint64
        L.orc_replay_build.restype = C.c_void_p
        L.orc_replay_build.argtypes = [C.c_uint64] * 4
        L.orc_replay_free.argtypes = [C.c_void_p]
        L.orc_replay_len.restype = C.c_uint64
        L.orc_replay_len.argtypes = [C.c_void_p]
        L.orc_replay_head.restype = C.c_uint64
        L.orc_replay_head.argtypes = [C.c_void_p]
        L.orc_replay_push.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 6
        L.orc_replay_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.orc_net_param_count.restype = C.c_int64
        L.orc_dqn_update.restype = C.c_float
        L.orc_dqn_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_adam_step.argtypes = [C.c_void_p] * 5 + [C.c_int64]
        L.orc_track.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int64]
        L.orc_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- RNG
class StdRng:
    """rand 0.8.5 StdRng (ChaCha12)."""

    def __init__(self, state: RngState):
        self.s = state

    @classmethod
    def seed_from_u64(cls, seed: int) -> "StdRng":
        s = RngState()
        lib().orc_rng_seed_from_u64(C.byref(s), C.c_uint64(seed))
        return cls(s)

    @classmethod
    def from_seed(cls, seed_bytes: bytes) -> "StdRng":
        assert len(seed_bytes) == 32
        s = RngState()
        lib().orc_rng_from_seed(C.byref(s), (C.c_uint8 * 32)(*seed_bytes))
        return cls(s)

    def next_u32(self) -> int:
        return lib().orc_rng_next_u32(C.byref(self.s))

    def next_u64(self) -> int:
        return lib().orc_rng_next_u64(C.byref(self.s))

    def sample_indices(self, size: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint64)
        lib().orc_sample_indices(C.byref(self.s), C.c_uint64(size), C.c_int(n), _p(out))
        return out


def seed_bytes_from_u64(seed: int) -> bytes:
    b = (C.c_uint8 * 32)()
    lib().orc_seed_bytes_from_u64(C.c_uint64(seed), b)
    return bytes(b)


def chacha_block(key_words, counter: int, rounds: int) -> np.ndarray:
    out = (C.c_uint32 * 16)()
    lib().orc_chacha_block((C.c_uint32 * 8)(*key_words), C.c_uint64(counter), C.c_int(rounds), out)
    return np.array(list(out), dtype=np.uint32)


# ----------------------------------------------------------------------------- replay
class Replay:
    """SimpleReplayBuffer restatement over opaque byte rows."""

    def __init__(self, capacity: int, seed: int, obs_bytes: int, act_bytes: int):
        self.h = lib().orc_replay_build(capacity, seed, obs_bytes, act_bytes)
        self.obs_bytes, self.act_bytes = obs_bytes, act_bytes

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_replay_free(self.h)
            self.h = None

    def __len__(self):
        return int(lib().orc_replay_len(self.h))

    @property
    def head(self):
        return int(lib().orc_replay_head(self.h))

    def push(self, obs, act, next_obs, reward, term, trunc):
        n = len(reward)
        obs = np.ascontiguousarray(obs).view(np.uint8).reshape(n, self.obs_bytes)
        next_obs = np.ascontiguousarray(next_obs).view(np.uint8).reshape(n, self.obs_bytes)
        act = np.ascontiguousarray(act).view(np.uint8).reshape(n, self.act_bytes)
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        term = np.ascontiguousarray(term, dtype=np.int8)
        trunc = np.ascontiguousarray(trunc, dtype=np.int8)
        lib().orc_replay_push(self.h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term), _p(trunc))

    def batch(self, n: int):
        ixs = np.empty(n, np.uint64)
        obs = np.empty((n, self.obs_bytes), np.uint8)
        nobs = np.empty((n, self.obs_bytes), np.uint8)
        act = np.empty((n, self.act_bytes), np.uint8)
        rew = np.empty(n, np.float32)
        term = np.empty(n, np.int8)
        trunc = np.empty(n, np.int8)
        rc = lib().orc_replay_batch(self.h, n, _p(ixs), _p(obs), _p(act), _p(nobs), _p(rew), _p(term), _p(trunc))
        if rc != 0:
            raise RuntimeError("batch() on an empty buffer")
        return dict(ixs=ixs, obs=obs, act=act, next_obs=nobs, reward=rew, is_terminated=term,
                    is_truncated=trunc)


# ----------------------------------------------------------------------------- nets / DQN
def cnn_cfg(out_dim: int, n_stack: int = 4) -> NetCfg:
    c = NetCfg()
    c.kind, c.n_stack, c.out_dim = 0, n_stack, out_dim
    return c


def mlp_cfg(in_dim: int, units, out_dim: int, activation_out: bool = False) -> NetCfg:
    c = NetCfg()
    c.kind, c.in_dim, c.n_units, c.out_dim = 1, in_dim, len(units), out_dim
    for i, u in enumerate(units):
        c.units[i] = u
    c.activation_out = int(activation_out)
    return c


def param_count(cfg: NetCfg) -> int:
    return int(lib().orc_net_param_count(C.byref(cfg)))


def net_forward(cfg: NetCfg, params: np.ndarray, x: np.ndarray) -> np.ndarray:
    B = x.shape[0]
    x = np.ascontiguousarray(x)
    out = np.empty((B, cfg.out_dim), np.float32)
    lib().orc_net_forward(C.byref(cfg), _p(params), _p(x), B, _p(out))
    return out


class DqnOracle:
    """Dqn::opt_ restatement (dqn/base.rs:182-200) over explicit minibatches."""

    def __init__(self, net: NetCfg, params: np.ndarray, *, lr: float, discount_factor=0.99,
                 double_dqn=False, critic_loss="Mse", clip_td_err=None, tau=0.005,
                 soft_update_interval=1):
        self.net = net
        self.q = np.array(params, dtype=np.float32, copy=True)
        self.q_tgt = self.q.copy()  # DqnModel::clone (dqn/model/base.rs:94-115)
        self.m = np.zeros_like(self.q)
        self.v = np.zeros_like(self.q)
        self.adam = AdamCfg(lr, 0.9, 0.999, 1e-8, 0)
        self.cfg = DqnCfg(discount_factor, int(double_dqn), {"Mse": 0, "SmoothL1": 1}[critic_loss],
                          int(clip_td_err is not None),
                          *(clip_td_err if clip_td_err is not None else (0.0, 0.0)))
        self.tau, self.soft_update_interval, self.soft_update_counter = tau, soft_update_interval, 0
        self.n_opts = 0

    def update(self, obs, act, next_obs, reward, term, weight=None, probe=False):
        B = len(reward)
        A = self.net.out_dim
        obs, next_obs = np.ascontiguousarray(obs), np.ascontiguousarray(next_obs)
        act = np.ascontiguousarray(act, dtype=np.int64).reshape(B)
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        term = np.ascontiguousarray(term, dtype=np.int8)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
        pr, bufs = None, {}
        if probe:
            bufs = dict(q_pred_all=np.empty((B, A), np.float32), q_next_all=np.empty((B, A), np.float32),
                        pred=np.empty(B, np.float32), tgt=np.empty(B, np.float32),
                        grads=np.empty_like(self.q), td_abs=np.empty(B, np.float32))
            pr = DqnProbe(*[bufs[n].ctypes.data for n, _ in DqnProbe._fields_])
        loss = lib().orc_dqn_update(C.byref(self.net), _p(self.q), _p(self.q_tgt), C.byref(self.adam),
                                    _p(self.m), _p(self.v), C.byref(self.cfg), B, _p(obs), _p(act),
                                    _p(next_obs), _p(reward), _p(term), _p(w),
                                    C.byref(pr) if pr is not None else None)
        # dqn/base.rs:190-196
        self.soft_update_counter += 1
        if self.soft_update_counter == self.soft_update_interval:
            self.soft_update_counter = 0
            lib().orc_track(_p(self.q_tgt), _p(self.q), C.c_double(self.tau), self.q.size)
        self.n_opts += 1
        bufs["loss"] = float(loss)
        return bufs


# ----------------------------------------------------------------------------- SAC
class SacCfg(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_pi_units", C.c_int32), ("pi_units", C.c_int32 * 8),
                ("n_q_units", C.c_int32), ("q_units", C.c_int32 * 8), ("n_critics", C.c_int32),
                ("gamma", C.c_double), ("tau", C.c_double), ("epsilon", C.c_double), ("min_lstd", C.c_double),
                ("max_lstd", C.c_double), ("reward_scale", C.c_double), ("critic_loss", C.c_int32),
                ("auto_alpha", C.c_int32), ("target_entropy", C.c_double)]


class SacRecord(C.Structure):
    _fields_ = [("loss_critic", C.c_float), ("loss_actor", C.c_float), ("ent_coef", C.c_float)]


class SacProbe(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pi_grads", "q_grads", "a", "log_p", "tgt", "q_pi", "q_pred", "q_next", "next_log_p", "next_a")]


class SacOracle:
    """Sac::opt_ restatement (sac/base.rs:175-198) over explicit minibatches and injected noise."""

    def __init__(self, obs_dim, act_dim, pi_units, q_units, pi_params, q_params_list, *, lr_actor, lr_critic, gamma=0.99,
                 tau=0.005, ent_coef=("Fix", 1.0), epsilon=1e-4, min_lstd=-20.0, max_lstd=2.0, reward_scale=1.0,
                 critic_loss="Mse"):
        import math
        L = lib()
        L.orc_sac_pi_param_count.restype = C.c_int64
        L.orc_sac_q_param_count.restype = C.c_int64
        c = SacCfg()
        c.obs_dim, c.act_dim, c.n_pi_units, c.n_q_units, c.n_critics = obs_dim, act_dim, len(pi_units), len(q_units), len(q_params_list)
        for i, u in enumerate(pi_units):
            c.pi_units[i] = u
        for i, u in enumerate(q_units):
            c.q_units[i] = u
        c.gamma, c.tau, c.epsilon, c.min_lstd, c.max_lstd, c.reward_scale = gamma, tau, epsilon, min_lstd, max_lstd, reward_scale
        c.critic_loss = {"Mse": 0, "SmoothL1": 1}[critic_loss]
        c.auto_alpha = int(ent_coef[0] == "Auto")
        c.target_entropy = ent_coef[1] if c.auto_alpha else 0.0
        self.cfg = c
        self.pi = np.array(pi_params, np.float32, copy=True)
        self.qs = [np.array(q, np.float32, copy=True) for q in q_params_list]
        self.qs_tgt = [q.copy() for q in self.qs]
        self.log_alpha = np.array([0.0 if c.auto_alpha else math.log(ent_coef[1])], np.float32)
        assert self.pi.size == L.orc_sac_pi_param_count(C.byref(c)) and self.qs[0].size == L.orc_sac_q_param_count(C.byref(c))
        self.pi_m, self.pi_v = np.zeros_like(self.pi), np.zeros_like(self.pi)
        self.q_m = [np.zeros_like(q) for q in self.qs]
        self.q_v = [np.zeros_like(q) for q in self.qs]
        self.al_m, self.al_v = np.zeros(1, np.float32), np.zeros(1, np.float32)
        self.adam_pi = AdamCfg(lr_actor, 0.9, 0.999, 1e-8, 0)
        self.adam_q = (AdamCfg * len(self.qs))(*[AdamCfg(lr_critic, 0.9, 0.999, 1e-8, 0) for _ in self.qs])
        self.adam_al = AdamCfg(ent_coef[2] if c.auto_alpha else 0.0, 0.9, 0.999, 1e-8, 0)

    def update(self, obs, act, next_obs, reward, term, z_actor, z_next, probe=True):
        f = lambda x: np.ascontiguousarray(x, dtype=np.float32)
        obs, act, next_obs, reward, z_actor, z_next = map(f, (obs, act, next_obs, reward, z_actor, z_next))
        term = np.ascontiguousarray(term, dtype=np.int8)
        B, NC = len(reward), len(self.qs)
        arr = lambda xs: (C.c_void_p * NC)(*[x.ctypes.data for x in xs])
        rec = SacRecord()
        bufs = dict(pi_grads=np.empty_like(self.pi), q_grads=np.empty((NC, self.qs[0].size), np.float32),
                    a=np.empty((B, self.cfg.act_dim), np.float32), log_p=np.empty(B, np.float32), tgt=np.empty(B, np.float32),
                    q_pi=np.empty((NC, B), np.float32), q_pred=np.empty((NC, B), np.float32), q_next=np.empty((NC, B), np.float32),
                    next_log_p=np.empty(B, np.float32), next_a=np.empty((B, self.cfg.act_dim), np.float32))
        pr = SacProbe(*[bufs[n].ctypes.data for n, _ in SacProbe._fields_])
        lib().orc_sac_update(C.byref(self.cfg), _p(self.pi), arr(self.qs), arr(self.qs_tgt), _p(self.log_alpha),
                             C.byref(self.adam_pi), _p(self.pi_m), _p(self.pi_v),
                             self.adam_q, arr(self.q_m), arr(self.q_v),
                             C.byref(self.adam_al), _p(self.al_m), _p(self.al_v),
                             C.c_int(B), _p(obs), _p(act), _p(next_obs), _p(reward), _p(term), _p(z_actor), _p(z_next),
                             C.byref(rec), C.byref(pr) if probe else None)
        bufs.update(loss_critic=rec.loss_critic, loss_actor=rec.loss_actor, ent_coef=rec.ent_coef)
        return bufs


# ----------------------------------------------------------------------------- IQN
class IqnCfg(C.Structure):
    _fields_ = [("psi_kind", C.c_int32), ("psi_in", C.c_int32), ("n_psi_units", C.c_int32), ("psi_units", C.c_int32 * 8),
                ("psi_activation_out", C.c_int32), ("feature_dim", C.c_int32), ("embed_dim", C.c_int32),
                ("n_f_units", C.c_int32), ("f_units", C.c_int32 * 8), ("n_actions", C.c_int32), ("discount_factor", C.c_double)]


class IqnProbe(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("z_pred", "z_tgt", "tgt", "grads")]


class IqnOracle:
    """Iqn::opt_ restatement (iqn/base.rs:172-191) over explicit minibatches and injected percent points."""

    def __init__(self, psi_kind, params, *, lr, feature_dim, embed_dim, f_units, n_actions, psi_in=0, psi_units=(),
                 psi_activation_out=True, discount_factor=0.99, tau=0.005, soft_update_interval=1):
        L = lib()
        L.orc_iqn_param_count.restype = C.c_int64
        L.orc_iqn_update.restype = C.c_float
        c = IqnCfg()
        c.psi_kind = 0 if psi_kind == "cnn" else 1
        c.psi_in, c.n_psi_units, c.psi_activation_out = psi_in, len(psi_units), int(psi_activation_out)
        for i, u in enumerate(psi_units):
            c.psi_units[i] = u
        c.feature_dim, c.embed_dim, c.n_f_units, c.n_actions, c.discount_factor = feature_dim, embed_dim, len(f_units), n_actions, discount_factor
        for i, u in enumerate(f_units):
            c.f_units[i] = u
        self.cfg = c
        self.p = np.array(params, np.float32, copy=True)
        assert self.p.size == L.orc_iqn_param_count(C.byref(c)), (self.p.size, L.orc_iqn_param_count(C.byref(c)))
        self.p_tgt = self.p.copy()
        self.m, self.v = np.zeros_like(self.p), np.zeros_like(self.p)
        self.adam = AdamCfg(lr, 0.9, 0.999, 1e-8, 0)
        self.tau, self.soft_update_interval, self.soft_update_counter = tau, soft_update_interval, 0

    def update(self, obs, act, next_obs, reward, term, tau_pred, tau_tgt):
        B = len(reward)
        A = self.cfg.n_actions
        obs, next_obs = np.ascontiguousarray(obs), np.ascontiguousarray(next_obs)
        act = np.ascontiguousarray(act, dtype=np.int64).reshape(B)
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        term = np.ascontiguousarray(term, dtype=np.int8)
        tp = np.ascontiguousarray(tau_pred, dtype=np.float32)
        tt = np.ascontiguousarray(tau_tgt, dtype=np.float32)
        bufs = dict(z_pred=np.empty((B, tp.shape[1], A), np.float32), z_tgt=np.empty((B, tt.shape[1], A), np.float32),
                    tgt=np.empty((B, tt.shape[1]), np.float32), grads=np.empty_like(self.p))
        pr = IqnProbe(*[bufs[n].ctypes.data for n, _ in IqnProbe._fields_])
        loss = lib().orc_iqn_update(C.byref(self.cfg), _p(self.p), _p(self.p_tgt), C.byref(self.adam), _p(self.m), _p(self.v),
                                    C.c_int(B), _p(obs), _p(act), _p(next_obs), _p(reward), _p(term), _p(tp), C.c_int(tp.shape[1]),
                                    _p(tt), C.c_int(tt.shape[1]), C.byref(pr))
        self.soft_update_counter += 1
        if self.soft_update_counter == self.soft_update_interval:
            self.soft_update_counter = 0
            lib().orc_track(_p(self.p_tgt), _p(self.p), C.c_double(self.tau), self.p.size)
        bufs["loss"] = float(loss)
        return bufs


class Explorer:
    """CPU restatement of Policy::sample's exploration (test infrastructure only).

    dqn/explorer.rs:29-31 (softmax -> multinomial), :68-90 (eps-greedy: one coin per CALL, eps linear in the
    number of calls), dqn/base.rs:229-236 (eval: 1 % uniformly random action), iqn/explorer.rs:28-30,78-96,
    iqn/base.rs:223 (eval: argmax).  The reference draws from fastrand's global *unseeded* generator, so its
    action stream is not reproducible; the build draws the same quantities, in the same order, from a
    seeded StdRng stream - which is what this class replays:
      f64  = (next_u64 >> 12) * 2^-52;  f32 = (next_u32 >> 9) * 2^-23;
      below(n) = Lemire multiply-shift with rejection (the algorithm behind fastrand::u32(..n)).
    """

    def __init__(self, kind: str = "softmax", eps_start=1.0, eps_final=0.02, final_step=100_000, n_opts=0, seed=0):
        self.kind, self.eps_start, self.eps_final, self.final_step, self.n_opts = kind, eps_start, eps_final, final_step, n_opts
        self.rng = StdRng.seed_from_u64(seed)
        self.n_samples_act = 0
        self.n_samples_best_act = 0

    def f64(self) -> float:
        return (self.rng.next_u64() >> 12) * (1.0 / 4503599627370496.0)

    def f32(self) -> np.float32:
        return np.float32(self.rng.next_u32() >> 9) * np.float32(1.0 / 8388608.0)

    def below(self, n: int) -> int:
        m = self.rng.next_u32() * n
        lo = m & 0xFFFFFFFF
        if lo < n:
            t = ((1 << 32) - n) % n
            while lo < t:
                m = self.rng.next_u32() * n
                lo = m & 0xFFFFFFFF
        return m >> 32

    def eps(self) -> float:
        d = (self.eps_start - self.eps_final) / float(self.final_step)
        return max(self.eps_start - d * float(self.n_opts), self.eps_final)

    def sample(self, q: np.ndarray, train: bool, dqn: bool = True):
        """q: [n_procs, A] float32 action values -> (actions int64 [n_procs], eps, is_random)."""
        import math
        q = np.asarray(q, np.float32)
        n, A = q.shape
        best = np.array([int(np.argmax(row)) for row in q], np.int64)   # first maximum
        if train:
            self.n_samples_act += 1
            if self.kind == "softmax":
                act = np.empty(n, np.int64)
                for i in range(n):
                    mx = np.max(q[i])
                    e = [math.exp(float(np.float32(v - mx))) for v in q[i]]
                    z = 0.0
                    for v in e:
                        z += v
                    u = self.f64() * z
                    c, a = 0.0, A - 1
                    for k in range(A):
                        c += e[k]
                        if c > u:
                            a = k
                            break
                    act[i] = a
                if np.array_equal(act, best):
                    self.n_samples_best_act += 1
                return act, 0.0, False
            eps = self.eps()
            is_random = self.f64() < eps
            self.n_opts += 1
            act = np.array([self.below(A) for _ in range(n)], np.int64) if is_random else best
            if np.array_equal(act, best):
                self.n_samples_best_act += 1
            return act, eps, is_random
        if dqn and self.f32() < np.float32(0.01):
            a = self.below(A)
            return np.full(n, a, np.int64), 0.0, True
        return best, 0.0, False


class SumTree:
    """sum_tree.rs restatement (f32, incremental sums)."""

    def __init__(self, capacity: int, alpha: float, normalize: str = "All"):
        L = lib()
        L.orc_sumtree_new.restype = C.c_void_p
        L.orc_sumtree_new.argtypes = [C.c_uint64, C.c_float, C.c_int]
        L.orc_sumtree_total.restype = C.c_float; L.orc_sumtree_total.argtypes = [C.c_void_p]
        L.orc_sumtree_max.restype = C.c_float; L.orc_sumtree_max.argtypes = [C.c_void_p]
        L.orc_sumtree_min_p.restype = C.c_float; L.orc_sumtree_min_p.argtypes = [C.c_void_p]
        L.orc_sumtree_n_samples.restype = C.c_uint64; L.orc_sumtree_n_samples.argtypes = [C.c_void_p]
        L.orc_sumtree_tree.restype = C.POINTER(C.c_float); L.orc_sumtree_tree.argtypes = [C.c_void_p]
        L.orc_sumtree_get.restype = C.c_uint64; L.orc_sumtree_get.argtypes = [C.c_void_p, C.c_float]
        L.orc_sumtree_update.restype = None; L.orc_sumtree_update.argtypes = [C.c_void_p, C.c_uint64, C.c_float]
        L.orc_sumtree_add.restype = None; L.orc_sumtree_add.argtypes = [C.c_void_p, C.c_uint64, C.c_float]
        L.orc_sumtree_free.restype = None; L.orc_sumtree_free.argtypes = [C.c_void_p]
        L.orc_sumtree_sample.restype = None
        L.orc_sumtree_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_iw_beta.restype = C.c_float; L.orc_iw_beta.argtypes = [C.c_float, C.c_float, C.c_uint64, C.c_uint64]
        self.capacity = capacity
        self.h = L.orc_sumtree_new(capacity, alpha, {"All": 0, "Batch": 1}[normalize])

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sumtree_free(self.h)
            self.h = None

    def add(self, ix, p): lib().orc_sumtree_add(self.h, ix, p)
    def update(self, ix, p): lib().orc_sumtree_update(self.h, ix, p)
    def get(self, s) -> int: return int(lib().orc_sumtree_get(self.h, s))
    def total(self) -> float: return float(lib().orc_sumtree_total(self.h))
    def max(self) -> float: return float(lib().orc_sumtree_max(self.h))
    def min_p(self) -> float: return float(lib().orc_sumtree_min_p(self.h))

    @property
    def n_samples(self) -> int: return int(lib().orc_sumtree_n_samples(self.h))

    def tree(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_sumtree_tree(self.h), shape=(2 * self.capacity - 1,)).copy()

    def sample(self, u: np.ndarray, beta: float):
        u = np.ascontiguousarray(u, np.float32)
        ixs = np.empty(len(u), np.int64)
        ws = np.empty(len(u), np.float32)
        lib().orc_sumtree_sample(self.h, len(u), beta, _p(u), _p(ixs), _p(ws))
        return ixs, ws


def iw_beta(beta_0, beta_final, n_opts_final, n_opts) -> float:
    SumTree  # noqa: B018  (argtypes are registered by the first SumTree)
    L = lib()
    L.orc_iw_beta.restype = C.c_float; L.orc_iw_beta.argtypes = [C.c_float, C.c_float, C.c_uint64, C.c_uint64]
    return float(L.orc_iw_beta(beta_0, beta_final, n_opts_final, n_opts))


class PerReplay:
    """SimpleReplayBuffer with `per_config: Some(..)` (generic_replay_buffer/base.rs:227-235, 295-316, 376-383,
    413-426): index/weight logic only (rows are gathered by the plain Replay restatement).  The batch's
    uniforms come from the buffer's StdRng as f32 = (next_u32 >> 9) * 2^-23 (the reference uses the unseeded
    fastrand::f32())."""

    def __init__(self, capacity, seed, alpha=0.6, beta_0=0.4, beta_final=1.0, n_opts_final=500_000, normalize="All"):
        self.capacity, self.i, self.size = capacity, 0, 0
        self.tree = SumTree(capacity, alpha, normalize)
        self.rng = StdRng.seed_from_u64(seed)
        self.beta_0, self.beta_final, self.n_opts_final, self.n_opts = beta_0, beta_final, n_opts_final, 0

    def push(self, length: int):
        max_p = self.tree.max()                      # set_priority: one max for the whole pushed block
        for j in range(length):
            self.tree.add((self.i + j) % self.capacity, max_p)
        self.i = (self.i + length) % self.capacity
        self.size = min(self.size + length, self.capacity)

    def beta(self) -> float:
        return iw_beta(self.beta_0, self.beta_final, self.n_opts_final, self.n_opts)

    def batch(self, n: int):
        u = np.array([np.float32(self.rng.next_u32() >> 9) * np.float32(1.0 / 8388608.0) for _ in range(n)], np.float32)
        return self.tree.sample(u, self.beta())

    def update_priority(self, ixs, td_errs):
        for ix, td in zip(ixs, td_errs):
            self.tree.update(int(ix), float(td))
        self.n_opts += 1


# ---- the optional device-native index generator (bdr_replay_config::index_rng = BDR_RNG_XOSHIRO256PP) ---------------------------------
# Not part of the reference (its stream is StdRng, above): xoshiro256++ 1.0 (Blackman / Vigna, public domain) restated, one generator
# per batch lane, lane j seeded with outputs 4j .. 4j+3 of SplitMix64(seed).  Test infrastructure like everything in oracle/.
_M64 = (1 << 64) - 1


def splitmix64_at(seed: int, n: int) -> int:
    z = (seed + (n + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class Xoshiro256pp:
    def __init__(self, s):
        self.s = [int(x) & _M64 for x in s]

    def next_u64(self) -> int:
        s = self.s
        sm = (s[0] + s[3]) & _M64
        result = ((((sm << 23) | (sm >> 41)) & _M64) + s[0]) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & _M64
        return result


class XoshiroLanes:
    """Index stream of a replay buffer built with index_rng = xoshiro256++: lane j draws sample j of every batch."""

    def __init__(self, seed: int):
        self.seed, self.lanes = seed, {}

    def sample_indices(self, size: int, n: int):
        import numpy as np
        out = np.empty(n, np.uint64)
        for j in range(n):
            if j not in self.lanes:
                self.lanes[j] = Xoshiro256pp([splitmix64_at(self.seed, 4 * j + i) for i in range(4)])
            out[j] = (self.lanes[j].next_u64() >> 32) % size
        return out
