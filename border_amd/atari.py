"""border-atari-env's frame preprocessing on the device (border-atari-env/src/env.rs:126-209, 263-324).

`AtariPreprocessor` keeps the `frames: [4][84][84]` stack of every environment in HBM; `reset` / `step` take the raw RGB
frames the emulator renders (`render_rgb24`) and return the stacked observation the reference's `BorderAtariObs` carries.
All arithmetic runs in the HIP library (`csrc/atari_prep.hip`).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib


class AtariPreprocessor:
    def __init__(self, n_envs: int, device: int = 0, width: int = 160, height: int = 210):
        self.n_envs, self.width, self.height = n_envs, width, height
        self._h = C.c_void_p()
        _lib.check(_lib.lib().bdr_atari_prep_create(device, n_envs, width, height, C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.lib().bdr_atari_prep_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _frames(self, f, n):
        f = np.ascontiguousarray(f, np.uint8)
        assert f.shape == (n, self.height, self.width, 3), f.shape
        return f

    def reset(self, env_ixs: Sequence[int], frames) -> np.ndarray:
        """Env::reset (env.rs:263-296) for the named environments; returns their observations [n][4][84][84]."""
        ix = np.ascontiguousarray(env_ixs, np.uint32)
        f = self._frames(frames, len(ix))
        _lib.check(_lib.lib().bdr_atari_prep_reset(self._h, len(ix), ix.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))
        return self.obs(ix)

    def step(self, env_ixs: Sequence[int], frames_a, frames_b) -> np.ndarray:
        """Env::step's observation path (env.rs:312-324): max of the two last frames, warp, grayscale, stack."""
        ix = np.ascontiguousarray(env_ixs, np.uint32)
        a, b = self._frames(frames_a, len(ix)), self._frames(frames_b, len(ix))
        _lib.check(_lib.lib().bdr_atari_prep_step(self._h, len(ix), ix.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                                  b.ctypes.data_as(C.c_void_p)))
        return self.obs(ix)

    def obs(self, env_ixs: Sequence[int]) -> np.ndarray:
        ix = np.ascontiguousarray(env_ixs, np.uint32)
        out = np.empty((len(ix), 4, 84, 84), np.uint8)
        _lib.check(_lib.lib().bdr_atari_prep_obs(self._h, len(ix), ix.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    ROW_BYTES = 4 * 84 * 84

    def device_stacks(self) -> int:
        """Device address of all current stacks, [n_envs][4][84][84] u8 (row stride ROW_BYTES): what `Dqn.sample_device` and the
        next_obs side of `SimpleReplayBuffer.push_device` take."""
        p = C.c_void_p()
        _lib.check(_lib.lib().bdr_atari_prep_device_stacks(self._h, C.byref(p)))
        return p.value

    def device_prev_stacks(self) -> int:
        """Device address of every environment's stack as it was before its last step (obs_t of the transition whose next_obs is
        `device_stacks()`)."""
        p = C.c_void_p()
        _lib.check(_lib.lib().bdr_atari_prep_device_prev_stacks(self._h, C.byref(p)))
        return p.value

    def reset_device(self, env_ixs: Sequence[int], frames) -> None:
        """`reset` without the read-back of the observations (they stay in HBM)."""
        ix = np.ascontiguousarray(env_ixs, np.uint32)
        f = self._frames(frames, len(ix))
        _lib.check(_lib.lib().bdr_atari_prep_reset(self._h, len(ix), ix.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))

    def step_device(self, env_ixs: Sequence[int], frames_a, frames_b) -> None:
        """`step` without the read-back of the observations."""
        ix = np.ascontiguousarray(env_ixs, np.uint32)
        a, b = self._frames(frames_a, len(ix)), self._frames(frames_b, len(ix))
        _lib.check(_lib.lib().bdr_atari_prep_step(self._h, len(ix), ix.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                                  b.ctypes.data_as(C.c_void_p)))

    @staticmethod
    def clip_reward(r: float, train: bool) -> float:
        """env.rs:159-169"""
        return float(_lib.lib().bdr_atari_clip_reward(C.c_float(r), int(bool(train))))


class AtariDeviceEnv:
    """One Atari-shaped environment whose observation never leaves HBM: an emulator (`reset() -> frame[H][W][3]`,
    `step(action) -> (frame_a, frame_b, reward, is_terminated, is_truncated)`: the two last frames of the skip-4 step,
    border-atari-env/src/env.rs:126-157) behind an `AtariPreprocessor`.  It speaks both conventions of the compiled loops:
      * host observations (`reset(None)`, `step_with_reset(act) -> Step`, as border's `Env`), and
      * device-resident observations (`device_obs = True`: `reset_into(dev_ptr)`, `step_into(act, obs_dev_ptr, init_dev_ptr)`,
        the callbacks of a `bdr_env_vtable` with `obs_on_device = 1`): the stack is copied inside the device."""

    def __init__(self, emulator, device: int = 0, device_obs: bool = True, train: bool = True):
        self.emu, self.device, self.device_obs, self.train = emulator, device, device_obs, train
        f = np.asarray(emulator.reset())
        self._first = f
        self.prep = AtariPreprocessor(1, device=device, width=f.shape[1], height=f.shape[0])

    def close(self):
        self.prep.close()

    # ---- device-resident convention
    def _copy(self, ptr):
        _lib.check(_lib.lib().bdr_atari_prep_copy_stack(self.prep._h, 0, C.c_void_p(ptr)))

    def _reset_frames(self):
        f = self._first if self._first is not None else np.asarray(self.emu.reset())
        self._first = None
        return f[None]

    def reset_into(self, ptr):
        self.prep.reset_device([0], self._reset_frames())
        self._copy(ptr)

    def step_into(self, act, obs_ptr, init_ptr):
        fa, fb, r, term, trunc = self.emu.step(int(np.asarray(act).ravel()[0]))
        self.prep.step_device([0], np.asarray(fa)[None], np.asarray(fb)[None])
        self._copy(obs_ptr)
        if term or trunc:
            self.prep.reset_device([0], self._reset_frames())
            self._copy(init_ptr)
        return self.clip_reward(r), int(term), int(trunc)

    def clip_reward(self, r):
        return AtariPreprocessor.clip_reward(r, self.train)

    # ---- host convention (border's Env): the same emulator calls, observations read back
    def reset(self, is_done=None):
        return self.prep.reset([0], self._reset_frames()).reshape(1, 4, 1, 84, 84)

    def step_with_reset(self, act):
        from .trainer import Step
        fa, fb, r, term, trunc = self.emu.step(int(np.asarray(act).ravel()[0]))
        obs = self.prep.step([0], np.asarray(fa)[None], np.asarray(fb)[None]).reshape(1, 4, 1, 84, 84)
        st = Step(np.asarray(act), obs, np.array([self.clip_reward(r)], np.float32), np.array([int(term)], np.int8), np.array([int(trunc)], np.int8))
        if st.is_done():
            st.init_obs = self.reset()
        return st
