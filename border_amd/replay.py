"""Host-side mirror of border-core's SimpleReplayBuffer over the C ABI.

Same names and argument meaning as the reference traits:
  ExperienceBufferBase::{push, len}   border-core/src/base/replay_buffer.rs:38-62
  ReplayBufferBase::{build, batch}    border-core/src/base/replay_buffer.rs:74-127
  SimpleReplayBufferConfig            border-core/src/generic_replay_buffer/config.rs:185-210
The ring, the StdRng (ChaCha12) index draw and the gather all run on the MI355X.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib


@dataclass
class PerConfig:
    """generic_replay_buffer/config.rs:45-83 (defaults alpha 0.6, beta 0.4 -> 1.0 over 500 000 opts, All)."""
    alpha: float = 0.6
    beta_0: float = 0.4
    beta_final: float = 1.0
    n_opts_final: int = 500_000
    normalize: str = "All"          # WeightNormalizer::{All, Batch} (sum_tree.rs:13-18)

    def to_c(self) -> "_lib.PerConfigC":
        c = _lib.PerConfigC()
        _lib.lib().bdr_per_config_default(C.byref(c))
        c.alpha, c.beta_0, c.beta_final, c.n_opts_final = self.alpha, self.beta_0, self.beta_final, self.n_opts_final
        c.normalize = {"All": 0, "Batch": 1}[self.normalize]
        return c


@dataclass
class SimpleReplayBufferConfig:
    """generic_replay_buffer/config.rs:185-210 (defaults: capacity 10000, seed 42, per None)."""
    capacity: int = 10000
    seed: int = 42
    per_config: Optional[object] = None
    # MI355X extension: store every distinct frame once instead of stacked obs + next_obs
    # (bdr_replay_config::frame_stack); frame_capacity 0 = capacity * 1.25 + 64 frames
    frame_stack: int = 0
    frame_capacity: int = 0
    # "StdRng" (default): the reference's index stream bit for bit.  "xoshiro256++": the device-native generator (one per batch lane
    # in HBM; bdr_replay_config::index_rng) - not the reference's stream, uniform sampling only
    index_rng: str = "StdRng"

    def capacity_(self, v):  # builder-style setters like the reference's
        self.capacity = v
        return self

    def seed_(self, v):
        self.seed = v
        return self


@dataclass
class GenericTransitionBatch:
    """generic_replay_buffer/batch.rs:89-162 (host copy of a sampled batch)."""
    obs: np.ndarray
    act: np.ndarray
    next_obs: np.ndarray
    reward: np.ndarray
    is_terminated: np.ndarray
    is_truncated: np.ndarray
    ix_sample: Optional[np.ndarray]
    weight: Optional[np.ndarray] = None

    def unpack(self):
        return (self.obs, self.act, self.next_obs, self.reward, self.is_terminated, self.is_truncated,
                self.ix_sample, self.weight)

    def __len__(self):
        return len(self.reward)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SimpleReplayBuffer:
    """SimpleReplayBuffer<O, A> with rows typed by (obs_shape, obs_dtype) / (act_shape, act_dtype)."""

    def __init__(self, config: SimpleReplayBufferConfig, obs_shape, obs_dtype, act_shape=(1,), act_dtype=np.int64,
                 device: int = 0):
        self.config = config
        self.device = device
        self.obs_shape, self.obs_dtype = tuple(obs_shape), np.dtype(obs_dtype)
        self.act_shape, self.act_dtype = tuple(act_shape), np.dtype(act_dtype)
        self.obs_bytes = int(np.prod(self.obs_shape)) * self.obs_dtype.itemsize
        self.act_bytes = int(np.prod(self.act_shape)) * self.act_dtype.itemsize
        cfg = _lib.ReplayConfig(config.capacity, config.seed, self.obs_bytes, self.act_bytes, device, config.frame_stack, config.frame_capacity,
                                 {"StdRng": 0, "xoshiro256++": 1}[config.index_rng], 0)
        h = C.c_void_p()
        _lib.check(_lib.lib().bdr_replay_create(C.byref(cfg), C.byref(h)))
        self._h = h
        if config.per_config is not None:      # base.rs:341-345: per_state = Some(PerState::new(..))
            _lib.check(_lib.lib().bdr_replay_enable_per(self._h, C.byref(config.per_config.to_c())))

    @classmethod
    def build(cls, config: SimpleReplayBufferConfig, **kw) -> "SimpleReplayBuffer":
        return cls(config, **kw)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().bdr_replay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ExperienceBufferBase -------------------------------------------------------------------
    def push(self, obs, act, next_obs, reward, is_terminated, is_truncated) -> None:
        reward = np.ascontiguousarray(reward, dtype=np.float32).reshape(-1)
        n = reward.shape[0]
        obs = np.ascontiguousarray(obs, dtype=self.obs_dtype).reshape(n, -1)
        next_obs = np.ascontiguousarray(next_obs, dtype=self.obs_dtype).reshape(n, -1)
        act = np.ascontiguousarray(act, dtype=self.act_dtype).reshape(n, -1)
        term = np.ascontiguousarray(is_terminated, dtype=np.int8).reshape(n)
        trunc = np.ascontiguousarray(is_truncated, dtype=np.int8).reshape(n)
        assert obs.nbytes == n * self.obs_bytes and act.nbytes == n * self.act_bytes
        _lib.check(_lib.lib().bdr_replay_push(self._h, n, _p(obs), _p(act), _p(next_obs), _p(reward), _p(term),
                                              _p(trunc)))

    def push_device(self, obs_dev: int, obs_stride: int, act, next_obs_dev: int, next_obs_stride: int, reward, is_terminated, is_truncated) -> None:
        """The same push for observation rows that already live in HBM (`bdr_replay_push_device`): obs_dev / next_obs_dev are device
        addresses (e.g. `AtariPreprocessor.device_prev_stacks()` / `.device_stacks()`), row k at address + k * stride; act / reward /
        flags are host arrays."""
        reward = np.ascontiguousarray(reward, dtype=np.float32).reshape(-1)
        n = reward.shape[0]
        act = np.ascontiguousarray(act, dtype=self.act_dtype).reshape(n, -1)
        term = np.ascontiguousarray(is_terminated, dtype=np.int8).reshape(n)
        trunc = np.ascontiguousarray(is_truncated, dtype=np.int8).reshape(n)
        assert act.nbytes == n * self.act_bytes
        _lib.check(_lib.lib().bdr_replay_push_device(self._h, n, C.c_void_p(obs_dev), obs_stride, _p(act), C.c_void_p(next_obs_dev), next_obs_stride,
                                                     _p(reward), _p(term), _p(trunc)))

    def len(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_replay_len(self._h, C.byref(n)))
        return n.value

    __len__ = len

    @property
    def head(self) -> int:
        n = C.c_uint64()
        _lib.check(_lib.lib().bdr_replay_head(self._h, C.byref(n)))
        return n.value

    def frames_used(self):
        """(frames allocated so far, frame capacity) of the single-frame store."""
        a, c = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().bdr_replay_frames_used(self._h, C.byref(a), C.byref(c)))
        return a.value, c.value

    # ReplayBufferBase -----------------------------------------------------------------------
    def batch(self, size: int) -> GenericTransitionBatch:
        ixs = np.empty(size, np.uint64)
        obs = np.empty((size,) + self.obs_shape, self.obs_dtype)
        nobs = np.empty((size,) + self.obs_shape, self.obs_dtype)
        act = np.empty((size,) + self.act_shape, self.act_dtype)
        rew = np.empty(size, np.float32)
        term = np.empty(size, np.int8)
        trunc = np.empty(size, np.int8)
        _lib.check(_lib.lib().bdr_replay_batch(self._h, size, _p(ixs), _p(obs), _p(act), _p(nobs), _p(rew), _p(term),
                                               _p(trunc)))
        weight = None
        if self.config.per_config is not None:  # base.rs:382: weight = Some(ws)
            weight = np.empty(size, np.float32)
            _lib.check(_lib.lib().bdr_replay_batch_weights(self._h, size, _p(weight)))
        return GenericTransitionBatch(obs, act, nobs, rew, term, trunc, ixs, weight)

    def sample_indices(self, size: int) -> np.ndarray:
        """The index draw of batch() alone (advances the RNG like batch(size))."""
        ixs = np.empty(size, np.uint64)
        _lib.check(_lib.lib().bdr_replay_sample_indices(self._h, size, _p(ixs)))
        return ixs

    def update_priority(self, ixs, td_errs) -> None:
        """base.rs:413-426: sum_tree.update(ix, td_err) in order + iw_scheduler.add_n_opts(); a no-op without PER."""
        if self.config.per_config is None:
            return None
        assert ixs is not None and td_errs is not None, "ixs / td_errs should be Some(_) in update_priority()"
        ixs = np.ascontiguousarray(ixs, np.uint64)
        td = np.ascontiguousarray(td_errs, np.float32)
        _lib.check(_lib.lib().bdr_replay_update_priority(self._h, len(ixs), _p(ixs), _p(td)))

    # PER introspection (parity tests) --------------------------------------------------------
    def per_info(self) -> dict:
        o = _lib.PerInfoC()
        _lib.check(_lib.lib().bdr_replay_per_info(self._h, C.byref(o)))
        return {k: getattr(o, k) for k in ("n_samples", "n_opts", "beta", "total", "max_p", "min_p")}

    def per_tree(self) -> np.ndarray:
        out = np.empty(2 * self.config.capacity - 1, np.float32)
        _lib.check(_lib.lib().bdr_replay_per_read(self._h, 0, _p(out), out.size))
        return out

    def per_get(self, s: float) -> int:
        ix = C.c_uint64()
        _lib.check(_lib.lib().bdr_replay_per_get(self._h, C.c_float(s), C.byref(ix)))
        return ix.value

    # benchmark / test helpers ---------------------------------------------------------------
    def fill_synthetic(self, n: int, seed: int = 0, kind: int = 0, n_actions: int = 6) -> None:
        _lib.check(_lib.lib().bdr_replay_fill_synthetic(self._h, n, seed, kind, n_actions))

    def read_rows(self, first: int, n: int):
        obs = np.empty((n,) + self.obs_shape, self.obs_dtype)
        nobs = np.empty((n,) + self.obs_shape, self.obs_dtype)
        act = np.empty((n,) + self.act_shape, self.act_dtype)
        rew = np.empty(n, np.float32)
        term = np.empty(n, np.int8)
        trunc = np.empty(n, np.int8)
        _lib.check(_lib.lib().bdr_replay_read_rows(self._h, first, n, _p(obs), _p(act), _p(nobs), _p(rew), _p(term),
                                                   _p(trunc)))
        return obs, act, nobs, rew, term, trunc
