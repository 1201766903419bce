"""Host restatement of the synthetic-fill generator of border_amd/csrc/replay.hip
(k_fill_synthetic, kind 0): lets tests rebuild the device ring bit for bit."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def synth_hash(seed, t, sec, w):
    with np.errstate(over="ignore"):
        x = np.uint64(seed + 1) * np.uint64(0x9E3779B97F4A7C15)
        x = x ^ ((np.asarray(t, np.uint64) + np.uint64(1)) * np.uint64(0xBF58476D1CE4E5B9))
        x = x ^ (((np.uint64(sec) << np.uint64(40)) | np.asarray(w, np.uint64)) * np.uint64(0x94D049BB133111EB))
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def atari_rows(seed, first, n, obs_bytes=28224, n_actions=6):
    """Returns obs[n,obs_bytes] u8, act[n] i64, next_obs, reward f32, term i8, trunc i8."""
    assert obs_bytes % 8 == 0
    t = np.arange(first, first + n, dtype=np.uint64)[:, None]
    w = np.arange(obs_bytes // 8, dtype=np.uint64)[None, :]
    obs = synth_hash(seed, t, 0, w).astype("<u8").view(np.uint8).reshape(n, obs_bytes)
    nobs = synth_hash(seed, t, 1, w).astype("<u8").view(np.uint8).reshape(n, obs_bytes)
    h = synth_hash(seed, t[:, 0], 2, 0)
    act = ((h & np.uint64(0xFFFFFFFF)) % np.uint64(n_actions)).astype(np.int64)
    u = (h >> np.uint64(40)).astype(np.int64)
    reward = np.where(u < 838861, -1.0, np.where(u < 15938355, 0.0, 1.0)).astype(np.float32)
    term = ((synth_hash(seed, t[:, 0], 2, 1) >> np.uint64(40)).astype(np.int64) < 83886).astype(np.int8)
    return obs, act, nobs, reward, term, np.zeros(n, np.int8)


def _normal(h):
    """synth_normal of replay.hip: Irwin-Hall(4) of the four 16-bit fields of the hash, centred, unit variance (f32)."""
    h = np.asarray(h, np.uint64)
    m = np.uint64(0xFFFF)
    u = ((h & m).astype(np.float32) + ((h >> np.uint64(16)) & m).astype(np.float32)
         + ((h >> np.uint64(32)) & m).astype(np.float32) + (h >> np.uint64(48)).astype(np.float32))
    return (u * np.float32(1.0 / 65536.0) - np.float32(2.0)) * np.float32(1.7320508)


def f32_rows(seed, first, n, obs_dim, act_dim):
    """k_fill_synthetic kind 1 with continuous actions (n_actions == 0): obs / next_obs ~ N(0,1) f32 [n, obs_dim],
    act ~ U(-1,1) f32 [n, act_dim], reward ~ N(0,1), P(term) = .005, trunc 0."""
    t = np.arange(first, first + n, dtype=np.uint64)[:, None]
    w = np.arange(obs_dim, dtype=np.uint64)[None, :]
    obs = _normal(synth_hash(seed, t, 0, w))
    nobs = _normal(synth_hash(seed, t, 1, w))
    g = synth_hash(seed, t, 3, np.arange(act_dim, dtype=np.uint64)[None, :])
    act = (g >> np.uint64(40)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
    reward = _normal(synth_hash(seed, t[:, 0], 2, 2))
    term = ((synth_hash(seed, t[:, 0], 2, 1) >> np.uint64(40)).astype(np.int64) < 83886).astype(np.int8)
    return obs, act, nobs, reward, term, np.zeros(n, np.int8)
