"""SimpleReplayBuffer restatement: ring wrap, size saturation, batch gather
(generic_replay_buffer/base.rs:295-316, 376-402)."""
import numpy as np
import pytest

from oracle import oracle as O


def _tr(rng, n, obs_bytes, act_bytes):
    return (rng.integers(0, 256, (n, obs_bytes), dtype=np.uint8), rng.integers(0, 256, (n, act_bytes), dtype=np.uint8),
            rng.integers(0, 256, (n, obs_bytes), dtype=np.uint8), rng.standard_normal(n).astype(np.float32),
            (rng.random(n) < .3).astype(np.int8), (rng.random(n) < .3).astype(np.int8))


def test_push_wraps_and_saturates():
    rng = np.random.default_rng(0)
    cap, ob, ab = 10, 12, 8
    r = O.Replay(cap, 42, ob, ab)
    mirror = dict(obs=np.zeros((cap, ob), np.uint8), rew=np.zeros(cap, np.float32))
    i = 0
    for n in [3, 4, 5, 7, 1]:
        t = _tr(rng, n, ob, ab)
        r.push(*t)
        for k in range(n):
            mirror["obs"][(i + k) % cap] = t[0][k]
            mirror["rew"][(i + k) % cap] = t[3][k]
        i = (i + n) % cap
        assert r.head == i
    assert len(r) == cap
    # batch: indices follow the StdRng stream, rows follow the mirror
    ref = O.StdRng.seed_from_u64(42)
    b = r.batch(6)
    assert (b["ixs"] == ref.sample_indices(cap, 6)).all()
    assert (b["obs"] == mirror["obs"][b["ixs"].astype(np.int64)]).all()
    assert (b["reward"] == mirror["rew"][b["ixs"].astype(np.int64)]).all()


def test_partial_fill_samples_only_filled_rows():
    rng = np.random.default_rng(1)
    r = O.Replay(100, 5, 4, 4)
    r.push(*_tr(rng, 7, 4, 4))
    for _ in range(20):
        assert r.batch(16)["ixs"].max() < 7


def test_empty_buffer_is_an_error():
    with pytest.raises(RuntimeError):
        O.Replay(4, 0, 4, 4).batch(1)
