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

    @staticmethod
    def clip_reward(r: float, train: bool) -> float:
        """env.rs:159-169"""
        return float(_lib.lib().bdr_atari_clip_reward(C.c_float(r), int(bool(train))))
