"""The device-resident actor path (SURVEY.md 8(f)-1 as written: `Policy::sample` "batched across many vectorised envs on device"):
observations that already live in HBM - the frame stacks of `bdr_atari_prep` (border-atari-env/src/env.rs:197-209, 312-324) -
go to `bdr_agent_sample_device` and `bdr_replay_push_device` without the HBM -> host -> HBM round trip of the host-pointer
calls (trainer/sampler.rs:99-144).  Bar: sampled actions, action values, ring rows, cursor, sampled indices and PER priorities
are BIT-identical to the host-pointer path on the same bytes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROW = 4 * 84 * 84


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _frames(rng, n, h=210, w=160):
    f = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    f[:, : h // 2] //= 3   # some structure, like a game screen
    return f


def _cnn(B, A=6, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=A), opt_config=B.OptimizerConfig.Adam(1e-4)),
                      device=0, batch_size=4, **kw)
    return B.Dqn.build(cfg)


@pytest.mark.parametrize("n_envs", [1, 7, 64])
def test_sample_device_equals_sample_of_the_same_rows_from_the_host(B, n_envs):
    from oracle import torch_ref as T
    rng = np.random.default_rng(n_envs)
    prep = B.AtariPreprocessor(n_envs)
    ixs = np.arange(n_envs)
    prep.reset_device(ixs, _frames(rng, n_envs))
    p0 = T.init_params(T.cnn_shapes(6), 5)
    host, dev = _cnn(B, train=True), _cnn(B, train=True)
    for a in (host, dev):
        a.set_params(p0, "qnet")
        a.set_explorer(B.EpsilonGreedy(final_step=30), seed=9)
    for call in range(40):
        prep.step_device(ixs, _frames(rng, n_envs), _frames(rng, n_envs))
        obs = prep.obs(ixs)                                                     # the same bytes on the host
        ah, ih = host.sample(obs, return_info=True)
        ad, idv = dev.sample_device(prep.device_stacks(), n_envs, ROW, return_info=True)
        assert ah.tolist() == ad.tolist() and ih == idv, call
        assert (host.qvalues(obs) == dev.qvalues_device(prep.device_stacks(), n_envs, ROW)).all()
    if n_envs >= 7:   # rows with a stride: every second environment
        m = (n_envs + 1) // 2
        q = dev.qvalues_device(prep.device_stacks(), m, 2 * ROW)
        assert (q == host.qvalues(prep.obs(ixs[::2]))).all()
    host.close(); dev.close(); prep.close()


def test_sample_device_for_mlp_iqn_and_sac_agents(B):
    """f32 rows in HBM: a replay buffer's device batch arrays play the producer."""
    rng = np.random.default_rng(3)
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=64, seed=1), (4,), np.float32)
    n = 40
    rows = rng.standard_normal((n, 4)).astype(np.float32)
    rb.push(rows, rng.integers(0, 3, (n, 1)), rows[::-1].copy(), np.zeros(n, np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))
    b = rb.batch(16)
    from border_amd import _lib
    db = _lib.DeviceBatch()
    _lib.check(_lib.lib().bdr_replay_last_batch(rb.handle, C.byref(db)))
    mk = lambda: B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=3),
                                                                          opt_config=B.OptimizerConfig.Adam(1e-3)), device=0, batch_size=8, train=True))
    host, dev = mk(), mk()
    for a in (host, dev):
        a.set_explorer(B.Softmax(), seed=4)
    for _ in range(20):
        assert host.sample(b.obs).tolist() == dev.sample_device(db.obs, 16, 16).tolist()
    assert (host.qvalues(b.obs) == dev.qvalues_device(db.obs, 16, 16)).all()
    host.close(); dev.close()
    # IQN (Mlp trunk), eval mode: argmax of the quantile-averaged values
    icfg = B.IqnConfig(f_config=B.MlpConfig(in_dim=4, units=(32,), out_dim=16, activation_out=True), feature_dim=16, embed_dim=8, m_units=(32,),
                       n_actions=3, lr=1e-3, batch_size=8, device=0, train=False)
    ih, idv = B.Iqn.build(icfg), B.Iqn.build(icfg)
    idv.set_params(ih.get_params("iqn"), "iqn")
    assert ih.sample(b.obs).tolist() == idv.sample_device(db.obs, 16, 16).tolist()
    ih.close(); idv.close()
    # SAC: eval-mode actions (mean of the squashed Gaussian) from device rows == from host rows
    scfg = B.SacConfig(obs_dim=4, act_dim=2, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=8, device=0, seed=1, train=False)
    sh, sd = B.Sac.build(scfg), B.Sac.build(scfg)
    assert (sh.sample(b.obs) == sd.sample_device(db.obs, 16, 16)).all()
    sh.close(); sd.close(); rb.close()


@pytest.mark.parametrize("per", [False, True])
def test_push_device_writes_the_ring_bit_for_bit_like_the_host_push(B, per):
    rng = np.random.default_rng(11)
    n_envs, cap = 6, 20
    prep = B.AtariPreprocessor(n_envs)
    ixs = np.arange(n_envs)
    prep.reset_device(ixs, _frames(rng, n_envs))
    mk = lambda: B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42, per_config=B.PerConfig() if per else None), (4, 1, 84, 84), np.uint8)
    rh, rd = mk(), mk()
    for step in range(9):   # 54 transitions into 20 slots: the ring wraps twice, pushes straddle the end
        obs = prep.obs(ixs)
        prep.step_device(ixs, _frames(rng, n_envs), _frames(rng, n_envs))
        nobs = prep.obs(ixs)
        act = rng.integers(0, 6, (n_envs, 1)).astype(np.int64)
        rew = rng.standard_normal(n_envs).astype(np.float32)
        term = (rng.random(n_envs) < 0.2).astype(np.int8); trunc = (rng.random(n_envs) < 0.1).astype(np.int8)
        if step % 3 == 2:   # a ragged push: every second environment, rows with a stride
            sl = slice(0, n_envs, 2)
            m = len(ixs[sl])
            rh.push(obs[sl], act[sl], nobs[sl], rew[sl], term[sl], trunc[sl])
            rd.push_device(prep.device_prev_stacks(), 2 * ROW, act[sl], prep.device_stacks(), 2 * ROW, rew[sl], term[sl], trunc[sl])
            assert m == 3
        else:
            rh.push(obs, act, nobs, rew, term, trunc)
            rd.push_device(prep.device_prev_stacks(), ROW, act, prep.device_stacks(), ROW, rew, term, trunc)
        assert len(rh) == len(rd) and rh.head == rd.head
        # the stacks before the step ARE the previous observation
        if step % 3 != 2:
            last = rd.read_rows((rd.head - n_envs) % cap, 1) if rd.head >= n_envs else None
            if last is not None:
                assert (last[0][0].reshape(4, 84, 84) == obs[0]).all() and (last[2][0].reshape(4, 84, 84) == nobs[0]).all()
    a, b = rh.read_rows(0, cap), rd.read_rows(0, cap)
    for x, y in zip(a, b):
        assert (x == y).all()
    for _ in range(5):
        bh, bd = rh.batch(8), rd.batch(8)
        assert bh.ix_sample.tolist() == bd.ix_sample.tolist() and (bh.obs == bd.obs).all() and (bh.next_obs == bd.next_obs).all()
        assert (bh.act == bd.act).all() and (bh.reward == bd.reward).all() and (bh.is_terminated == bd.is_terminated).all()
        if per:
            assert (bh.weight == bd.weight).all()
    rh.close(); rd.close(); prep.close()


def test_device_calls_refuse_host_pointers_and_the_single_frame_store(B):
    from border_amd._lib import BdrError
    prep = B.AtariPreprocessor(2)
    host_rows = np.zeros((2, 4, 84, 84), np.uint8)
    a = _cnn(B)
    with pytest.raises(BdrError, match="not device memory"):
        a.sample_device(host_rows.ctypes.data, 2, ROW)
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=8, seed=1), (4, 1, 84, 84), np.uint8)
    args = (np.zeros((2, 1), np.int64),)
    with pytest.raises(BdrError, match="device memory"):
        rb.push_device(host_rows.ctypes.data, ROW, args[0], prep.device_stacks(), ROW, np.zeros(2, np.float32), np.zeros(2, np.int8), np.zeros(2, np.int8))
    fr = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=8, seed=1, frame_stack=4), (4, 1, 84, 84), np.uint8)
    with pytest.raises(BdrError, match="single-frame store"):
        fr.push_device(prep.device_prev_stacks(), ROW, args[0], prep.device_stacks(), ROW, np.zeros(2, np.float32), np.zeros(2, np.int8), np.zeros(2, np.int8))
    assert len(rb) == 0 and len(fr) == 0
    a.close(); rb.close(); fr.close(); prep.close()


class _Emulator:
    """Seeded stand-in for the ALE: small RGB frames, rewards and episode ends independent of the action (like SyntheticEnv)."""

    def __init__(self, seed, h=84, w=96, p_term=0.08):
        self.rng, self.h, self.w, self.p = np.random.default_rng(seed), h, w, p_term

    def _f(self):
        return self.rng.integers(0, 256, (self.h, self.w, 3), dtype=np.uint8)

    def reset(self):
        return self._f()

    def step(self, action):
        fa, fb = self._f(), self._f()
        r = float(self.rng.choice([-2.0, 0.0, 3.0], p=[0.1, 0.8, 0.1]))
        return fa, fb, r, bool(self.rng.random() < self.p), bool(self.rng.random() < 0.02)


def test_compiled_trainer_with_device_resident_observations_equals_the_host_observation_run(B):
    """bdr_trainer_train with an environment whose observations stay in HBM (bdr_env_vtable::obs_on_device: AtariDeviceEnv copies its
    frame stack inside the device; the loop acts through bdr_agent_sample_device and pushes through bdr_replay_push_device) against
    the same run with host observations: the ring, the index stream and the trained parameters must be bit-identical, the episode
    and step counters equal (trainer/sampler.rs:99-144, step_proc.rs:103-137 - incl. init_obs after terminal steps)."""
    from oracle import torch_ref as T
    out = []
    for device_obs in (False, True):
        env = B.AtariDeviceEnv(_Emulator(5), device_obs=device_obs)
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=48, seed=7), (4, 1, 84, 84), np.uint8)
        a = _cnn(B, train=True)
        a.set_params(T.init_params(T.cnn_shapes(6), 3), "qnet"); a.set_params(T.init_params(T.cnn_shapes(6), 3), "qnet_tgt")
        a.set_explorer(B.EpsilonGreedy(final_step=40), seed=2)
        tr = B.NativeTrainer(B.TrainerConfig(max_opts=30, opt_interval=2, warmup_period=8))
        stats = tr.train(env, a, rb, (4, 1, 84, 84), np.uint8)
        a.sync()
        out.append((stats["env_steps"], stats["opt_steps"], stats["n_episodes"], rb.read_rows(0, 48), rb.sample_indices(8).tolist(),
                    a.get_params("qnet"), a.get_params("qnet_tgt")))
        a.close(); rb.close(); env.close()
    h, d = out
    assert h[:3] == d[:3] and h[2] >= 2 and h[0] == 66          # wrapped ring, several episode ends
    for x, y in zip(h[3], d[3]):
        assert (x == y).all()
    assert h[4] == d[4] and (h[5] == d[5]).all() and (h[6] == d[6]).all()


def test_async_trainer_with_device_resident_actors_moves_the_same_transitions(B):
    """bdr_async_train with device-resident actors: the actors' messages carry device rows (allocated per message, pushed with
    buffer_push_device by the learner, then freed).  Thread interleaving is not reproducible, so the property checked is per actor:
    the chain of transitions each actor produced - found in the learner's ring - is exactly what that actor's emulator and
    explorer stream produce in the host-observation run (same seeds, actors never adopt a newer model: sync_interval > max_opts)."""
    from oracle import torch_ref as T
    rows = {}
    for device_obs in (False, True):
        envs = [B.AtariDeviceEnv(_Emulator(20 + i, p_term=0.1), device_obs=device_obs) for i in range(2)]
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=400, seed=7), (4, 1, 84, 84), np.uint8)
        learner = _cnn(B, train=True)
        p0 = T.init_params(T.cnn_shapes(6), 3)
        learner.set_params(p0, "qnet"); learner.set_params(p0, "qnet_tgt")
        actors = []
        for i in range(2):
            ag = _cnn(B, train=True)
            ag.set_explorer(B.EpsilonGreedy(final_step=50), seed=30 + i)
            actors.append(ag)
        tr = B.AsyncTrainer(B.AsyncTrainerConfig(max_opts=6, warmup_period=40, sync_interval=1000, record_agent_info_interval=0, record_compute_cost_interval=0,
                                                 warmup_sleep_ms=1), B.ActorManagerConfig(n_buffer=5))
        pushed = []
        st = tr.train(learner, rb, actors, envs, (4, 1, 84, 84), np.uint8, on_event=lambda actor, a, b, ev, v: pushed.append((actor, v)) if ev == "push" else None)
        assert st.opt_steps == 6 and st.samples_total >= 40
        n = min(len(rb), 400)
        obs, act, nobs, rew, term, trunc = rb.read_rows(0, n)
        # the ring in push order: message k of `pushed` covers the next v rows and belongs to one actor
        per_actor, o = {0: [], 1: []}, 0
        for actor, v in pushed:
            if o + v > n:
                break
            per_actor[actor].append((obs[o:o + v].copy(), act[o:o + v].copy(), nobs[o:o + v].copy(), rew[o:o + v].copy(), term[o:o + v].copy(), trunc[o:o + v].copy()))
            o += v
        rows[device_obs] = {k: [np.concatenate([m[j] for m in v]) for j in range(6)] for k, v in per_actor.items() if v}
        for ag in actors: ag.close()
        learner.close(); rb.close()
        for e in envs: e.close()
    for actor in (0, 1):
        assert actor in rows[False] and actor in rows[True]
        m = min(len(rows[False][actor][3]), len(rows[True][actor][3]))
        assert m >= 10
        for x, y in zip(rows[False][actor], rows[True][actor]):
            assert (x[:m] == y[:m]).all()


def test_async_trainer_refuses_device_rows_of_another_gpu_before_it_starts(B):
    """An actor whose device-resident observations live on another GPU than the learner's ring cannot push them without a host copy
    (bdr_replay_push_device): bdr_async_train says so before any actor thread starts, not at the first message of a running job."""
    from border_amd._lib import BdrError
    envs = [B.AtariDeviceEnv(_Emulator(20 + i, p_term=0.1), device_obs=True) for i in range(2)]
    envs[1].device = 1        # what the actor's function table will carry; the check comes before any use of that GPU
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100, seed=7), (4, 1, 84, 84), np.uint8)
    learner = _cnn(B, train=True)
    actors = [_cnn(B, train=True) for _ in range(2)]
    tr = B.AsyncTrainer(B.AsyncTrainerConfig(max_opts=2, warmup_period=10, sync_interval=1000, record_agent_info_interval=0, record_compute_cost_interval=0,
                                             warmup_sleep_ms=1), B.ActorManagerConfig(n_buffer=5))
    pushed = []
    with pytest.raises(BdrError, match="actor 1 keeps its observations on GPU 1"):
        tr.train(learner, rb, actors, envs, (4, 1, 84, 84), np.uint8, on_event=lambda actor, a, b, ev, v: pushed.append(v) if ev == "push" else None)
    assert not pushed and len(rb) == 0 and learner.n_opts == 0
    for ag in actors: ag.close()
    learner.close(); rb.close()
    for e in envs: e.close()
