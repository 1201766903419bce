"""SURVEY.md 8(f) rank 4, second half: the single-frame store (bdr_replay_config::frame_stack).  border-atari-env keeps a
newest-first stack of four 84x84 frames (env.rs:197-209 stack_frame; reset fills all four slots with the first frame,
:263-296), so obs_t and next_obs_t share three frames and next_obs_t is obs_t+1 inside an episode: the stacked ring stores
every frame eight times.  The frame store keeps each distinct frame once and rebuilds the stacks in the gather.

Contract under test: for the SAME pushes the frame store and the plain ring are indistinguishable - len / head, the index
stream, every field of every batch, every row - across episode boundaries, ragged pushes and ring wrap; input that has no
frame structure at all is still stored exactly; running out of frames is a loud error, never silent corruption."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SHAPE = (4, 1, 84, 84)


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


class AtariLikeStream:
    """Transitions the way BorderAtariEnv + SimpleStepProcessor produce them: a stack of 4 frames, newest first; a step shifts
    one new frame in; after a terminal step the next transition starts from a reset stack (4 copies of the first frame)."""

    def __init__(self, seed, p_done=0.08):
        self.rng = np.random.default_rng(seed)
        self.p_done = p_done
        self.stack = None
        self.episodes = 0

    def frame(self):
        return self.rng.integers(0, 256, (1, 84, 84), dtype=np.uint8)

    def take(self, n):
        obs, nxt, act, rew, term = [], [], [], [], []
        for _ in range(n):
            if self.stack is None:
                f = self.frame()
                self.stack = np.stack([f] * 4)                   # reset: all four slots
                self.episodes += 1
            o = self.stack.copy()
            self.stack = np.concatenate([self.frame()[None], self.stack[:3]])   # stack_frame: newest first
            done = self.rng.random() < self.p_done
            obs.append(o); nxt.append(self.stack.copy())
            act.append(self.rng.integers(0, 6)); rew.append(self.rng.standard_normal()); term.append(1 if done else 0)
            if done:
                self.stack = None
        n_ = len(act)
        return (np.stack(obs), np.array(act, np.int64).reshape(n_, 1), np.stack(nxt), np.array(rew, np.float32), np.array(term, np.int8),
                np.zeros(n_, np.int8))


def same_batch(g, w, bs):
    assert (g.ix_sample == w.ix_sample).all()
    assert (g.obs == w.obs).all() and (g.next_obs == w.next_obs).all()
    assert (g.act == w.act).all() and (g.reward.view(np.uint32) == w.reward.view(np.uint32)).all()
    assert (g.is_terminated == w.is_terminated).all() and (g.is_truncated == w.is_truncated).all()


def test_frame_store_equals_the_stacked_ring_for_the_same_pushes(B):
    cap = 150
    plain = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), SHAPE, np.uint8)
    fr = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42, frame_stack=4), SHAPE, np.uint8)
    st = AtariLikeStream(1)
    total = 0
    for n in [1, 1, 7, 30, 1, 64, 3, 100, 2, 149, 1, 40]:       # ragged, crosses episode ends, wraps the ring twice
        tr = st.take(n)
        plain.push(*tr); fr.push(*tr)
        total += n
        assert len(plain) == len(fr) == min(total, cap) and plain.head == fr.head
        for bs in (1, 33):
            same_batch(fr.batch(bs), plain.batch(bs), bs)
        k = min(len(fr), 9)
        a, b = fr.read_rows(0, k), plain.read_rows(0, k)
        assert all((x == y).all() for x, y in zip(a, b))
    used, fcap = fr.frames_used()
    # one new frame per step + ONE per reset stack (its four slots hold the same frame, stored once)
    assert used == total + st.episodes and fcap == cap + cap // 4 + 64
    plain.close(); fr.close()


def test_unstructured_rows_are_stored_exactly_and_exhaustion_is_loud(B):
    rng = np.random.default_rng(3)
    cap = 40

    def rows(n):
        return (rng.integers(0, 256, (n,) + SHAPE, dtype=np.uint8), rng.integers(0, 6, (n, 1)).astype(np.int64),
                rng.integers(0, 256, (n,) + SHAPE, dtype=np.uint8), rng.standard_normal(n).astype(np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))
    plain = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=7), SHAPE, np.uint8)
    fr = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=7, frame_stack=4, frame_capacity=8 * (cap + 4)), SHAPE, np.uint8)
    for n in (5, 37, 20, 40, 11):                                # nothing shares a frame: 8 frames per transition, wrapping
        tr = rows(n)
        plain.push(*tr); fr.push(*tr)
        same_batch(fr.batch(16), plain.batch(16), 16)
    assert fr.frames_used()[0] == 8 * (5 + 37 + 20 + 40 + 11)
    plain.close(); fr.close()
    # too small a store for unstructured rows: the push that would overwrite a live frame fails, what was accepted stays intact
    small = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=7, frame_stack=4, frame_capacity=100), SHAPE, np.uint8)
    ref = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=7), SHAPE, np.uint8)
    tr = rows(30)
    with pytest.raises(B.BdrError) as e:
        small.push(*tr)
    assert e.value.code == 1 and "frame" in str(e.value)
    k = len(small)
    assert 0 < k < 30                                            # 100 frames hold 12 unshared transitions
    ref.push(*[x[:k] for x in tr])
    same_batch(small.batch(8), ref.batch(8), 8)
    small.close(); ref.close()


def test_device_fill_and_dqn_opt_are_identical_on_both_stores(B):
    """The device fill of the frame store (one continuous episode) against its host restatement, and ten DQN opt steps over it
    against ten steps over a stacked ring holding the same rows: parameters bit-identical (the agent only ever sees the batch)."""
    from tests import synth
    n, A = 300, 6
    fr = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=n, seed=42, frame_stack=4), SHAPE, np.uint8)
    fr.fill_synthetic(n, seed=9, kind=0, n_actions=A)
    w = np.arange(7056 // 8, dtype=np.uint64)[None, :]
    frames = synth.synth_hash(9, np.arange(n + 4, dtype=np.uint64)[:, None], 7, w).astype("<u8").view(np.uint8).reshape(n + 4, 1, 84, 84)
    obs = np.stack([np.stack([frames[t + 3 - j] for j in range(4)]) for t in range(n)])
    nxt = np.stack([np.stack([frames[t + 4 - j] for j in range(4)]) for t in range(n)])
    e = synth.atari_rows(9, 0, n)
    o, a, x, r, t, u = fr.read_rows(0, n)
    assert (o == obs).all() and (x == nxt).all() and (a[:, 0] == e[1]).all() and (r == e[3]).all() and (t == e[4]).all()
    plain = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=n, seed=42), SHAPE, np.uint8)
    plain.push(obs, e[1].reshape(-1, 1), nxt, e[3], e[4], e[5])
    # a push after the device fill continues the episode: one more frame, not five
    used0 = fr.frames_used()[0]
    more = (nxt[-1:], np.array([[1]]), np.concatenate([frames[:1][None], nxt[-1:, :3]], 1), np.array([0.5], np.float32), np.array([0], np.int8), np.array([0], np.int8))
    fr.push(*more); plain.push(*more)
    assert fr.frames_used()[0] == used0 + 1
    outs = []
    for rb in (plain, fr):
        cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=A), opt_config=B.OptimizerConfig.Adam(1e-4)),
                          device=0, batch_size=32, critic_loss="SmoothL1", tau=1.0, soft_update_interval=4, param_seed=3)
        ag = B.Dqn.build(cfg)
        for _ in range(10):
            ag.opt(rb)
        ag.sync()
        outs.append(ag.get_params("qnet"))
        ag.close()
    assert (outs[0] == outs[1]).all()
    plain.close(); fr.close()


def test_one_million_transition_frame_store(B):
    """BASELINE-size ring as a frame store: 1 000 000 transitions in 8.8 GB of frames + 48 MB of records (the stacked ring: 56.6
    GB).  Index stream identical to the CPU restatement of batch(); gathered stacks are the frames the records name."""
    from oracle import oracle as O
    from tests import synth
    cap, bs = 1_000_000, 256
    fr = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42, frame_stack=4), SHAPE, np.uint8)
    fr.fill_synthetic(cap, seed=0, kind=0, n_actions=6)
    assert fr.frames_used() == (cap + 4, cap + cap // 4 + 64) and len(fr) == cap
    ref = O.StdRng.seed_from_u64(42)
    w = np.arange(7056 // 8, dtype=np.uint64)[None, :]
    for it in range(200):
        want = ref.sample_indices(cap, bs)
        if it % 50 == 0:
            g = fr.batch(bs)
            assert (g.ix_sample == want).all()
            for k in (0, 77, 255):
                t = int(want[k])
                q = np.array([t + 3, t + 2, t + 1, t], np.uint64)[:, None]
                assert (g.obs[k].reshape(4, -1) == synth.synth_hash(0, q, 7, w).astype("<u8").view(np.uint8).reshape(4, -1)).all()
                assert (g.next_obs[k].reshape(4, -1) == synth.synth_hash(0, q + np.uint64(1), 7, w).astype("<u8").view(np.uint8).reshape(4, -1)).all()
        else:
            assert (fr.sample_indices(bs) == want).all()
    fr.close()
