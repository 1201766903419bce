"""Policy::sample through the C ABI (device forward + exploration) vs the CPU restatement.

Row (f)-1 of the scope table: dqn/base.rs:211-242, dqn/explorer.rs:29-31,68-90, iqn/base.rs:204-228.  The
library and oracle/oracle.py::Explorer draw the same quantities in the same order from the same seeded
ChaCha12 stream, so the action sequences must agree exactly (integer parity), given the Q-values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _cnn_agent(B, A=6, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=A),
                                                    opt_config=B.OptimizerConfig.Adam(1e-4)), device=0, batch_size=4, **kw)
    return B.Dqn.build(cfg)


def _mlp_agent(B, A=3, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=A),
                                                    opt_config=B.OptimizerConfig.Adam(1e-3)), device=0, batch_size=32, **kw)
    return B.Dqn.build(cfg)


def test_eps_greedy_sequence_matches_oracle_exactly(B):
    from oracle.oracle import Explorer
    rng = np.random.default_rng(0)
    a = _mlp_agent(B, train=True)
    a.set_explorer(B.EpsilonGreedy(final_step=50), seed=7)
    ref = Explorer("eps_greedy", final_step=50, seed=7)
    for call in range(300):
        obs = rng.standard_normal((5, 4)).astype(np.float32)
        q = a.qvalues(obs)
        act, info = a.sample(obs, return_info=True)
        ract, reps, rrand = ref.sample(q, train=True)
        assert act.tolist() == ract.tolist(), call
        assert info["is_random"] == rrand and abs(info["eps"] - reps) < 1e-15
    assert info["n_samples_act"] == 300 == ref.n_samples_act
    assert info["n_samples_best_act"] == ref.n_samples_best_act
    assert a.explorer_state()["n_opts"] == 300
    a.close()


def test_softmax_sequence_matches_oracle_and_softmax_law(B):
    from oracle.oracle import Explorer
    rng = np.random.default_rng(1)
    a = _mlp_agent(B, train=True)
    a.set_explorer(B.Softmax(), seed=3)
    ref = Explorer("softmax", seed=3)
    obs = rng.standard_normal((4, 4)).astype(np.float32)
    q = a.qvalues(obs)
    acts = []
    for call in range(1500):
        act = a.sample(obs)
        ract, _, _ = ref.sample(q, train=True)
        assert act.tolist() == ract.tolist(), call
        acts.append(act)
    acts = np.stack(acts)
    for r in range(4):
        p = np.exp(q[r] - q[r].max()); p /= p.sum()
        freq = np.bincount(acts[:, r], minlength=3) / len(acts)
        assert np.abs(freq - p).max() < 0.05
    a.close()


@pytest.mark.parametrize("A", [4, 9, 18, 33])
def test_eps_greedy_sequence_of_the_cnn_agent_matches_oracle_exactly(B, A):
    """The Nature-CNN agent's Policy::sample at action counts in each register block of its head kernel (8 / 24 / 64;
    Atari's minimal action sets have 3...18 actions, border-atari-env/src/env.rs:97-103): argmax rows, random rows, the
    epsilon schedule and the counters against the CPU restatement, exactly."""
    from oracle.oracle import Explorer
    from oracle import torch_ref as T
    rng = np.random.default_rng(A)
    a = _cnn_agent(B, A=A, train=True)
    a.set_params(T.init_params(T.cnn_shapes(A), 5), "qnet")
    a.set_explorer(B.EpsilonGreedy(final_step=40), seed=11 + A)
    ref = Explorer("eps_greedy", final_step=40, seed=11 + A)
    seen = set()
    for call in range(120):
        obs = rng.integers(0, 256, (3, 4, 1, 84, 84), dtype=np.uint8)
        q = a.qvalues(obs)
        assert q.shape == (3, A)
        act, info = a.sample(obs, return_info=True)
        ract, reps, rrand = ref.sample(q, train=True)
        assert act.tolist() == ract.tolist(), call
        assert info["is_random"] == rrand and abs(info["eps"] - reps) < 1e-15
        seen.update(act.tolist())
    assert info["n_samples_act"] == 120 == ref.n_samples_act and info["n_samples_best_act"] == ref.n_samples_best_act
    assert max(seen) < A and len(seen) > 1
    a.close()


def test_eval_mode_cnn_greedy_with_one_percent_random(B):
    from oracle.oracle import Explorer
    from oracle import torch_ref as T
    rng = np.random.default_rng(2)
    a = _cnn_agent(B, train=False)
    a.set_params(T.init_params(T.cnn_shapes(6), 5), "qnet")
    a.set_explorer(B.Softmax(), seed=21)
    ref = Explorer("softmax", seed=21)
    n_rand = 0
    for call in range(400):
        obs = rng.integers(0, 256, (2, 4, 1, 84, 84), dtype=np.uint8)
        q = a.qvalues(obs)
        act, info = a.sample(obs, return_info=True)
        ract, _, rrand = ref.sample(q, train=False, dqn=True)
        assert act.tolist() == ract.tolist() and info["is_random"] == rrand
        if not rrand:
            assert act.tolist() == q.argmax(1).tolist()
        n_rand += rrand
    assert info["n_samples_act"] == 0           # only counted in train mode (dqn/base.rs:215)
    assert n_rand <= 15
    a.close()


def test_iqn_sample_eval_is_argmax_and_train_explores(B):
    from oracle.oracle import Explorer
    rng = np.random.default_rng(3)
    f_cfg = B.MlpConfig(in_dim=4, units=(32,), out_dim=16)
    cfg = B.IqnConfig(f_config=f_cfg, feature_dim=16, embed_dim=8, m_units=(32,), n_actions=3, lr=1e-3, batch_size=8, device=0, train=False)
    a = B.Iqn(cfg)
    obs = rng.standard_normal((6, 4)).astype(np.float32)
    q = a.qvalues(obs)     # Const32 percent points: deterministic
    for _ in range(20):
        assert a.sample(obs).tolist() == q.argmax(1).tolist()
    a.train()
    a.set_explorer(B.EpsilonGreedy(final_step=10), seed=4)
    ref = Explorer("eps_greedy", final_step=10, seed=4)
    for call in range(100):
        act = a.sample(obs)
        ract, _, _ = ref.sample(q, train=True, dqn=False)
        assert act.tolist() == ract.tolist(), call
    a.close()


@pytest.mark.parametrize("A", [6, 18, 33])
def test_acting_kernels_agree_with_the_training_forward(B, A):
    """Acting-sized calls (n <= 8 rows) take their own kernels (csrc/act_small.hpp: 32 x 32 tiles x k-slices, the last workgroup of a
    tile adds the slices, l2 and the hand-over to the host by the workgroup that finishes l1); larger calls run the training forward.
    The same rows through both: the acting kernels multiply in exact f32, the training forward's conv2 / conv3 on the bf16 matrix cores keep
    six of the nine partial products of their split operands (~2e-6 per layer; BDR_DQN_F32_EXACT=1: exact, 1e-6 between the two) -
    5e-6 relative at most, and the same greedy action wherever the two leading Q-values are not within that distance of each other."""
    rng = np.random.default_rng(A)
    from oracle import torch_ref as T
    a = _cnn_agent(B, A=A, train=False)
    a.set_params(T.init_params(T.cnn_shapes(A), 9), "qnet")
    for n in (1, 2, 3, 5, 8):
        obs = rng.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8)
        q_act = a.qvalues(obs)                                                       # acting kernels
        filler = rng.integers(0, 256, (24, 4, 1, 84, 84), dtype=np.uint8)
        q_train = a.qvalues(np.concatenate([obs, filler]))[:n]                       # the training forward (32 rows)
        scale = np.abs(q_train).max()
        assert np.abs(q_act - q_train).max() <= 5e-6 * scale, (n, np.abs(q_act - q_train).max() / scale)
        top2 = np.sort(q_train, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 2e-5 * scale
        assert (q_act.argmax(1)[clear] == q_train.argmax(1)[clear]).all()
    # The boundary between the two families is n = 8 / 9 (include/border_amd.h at bdr_agent_sample): the SAME 8 rows alone and with a ninth.
    # Near-ties are engineered in the output layer: action 1's row = action 0's row (an exact tie: the first maximum on both paths), action
    # 2's = action 0's with the bias raised by 1e-4 of |Q| (a gap well above the 2e-5 band: action 2 wherever the three lead).
    p = a.get_params("qnet").copy()
    shapes = T.cnn_shapes(A)
    off = np.cumsum([0] + [int(np.prod(sh)) for sh in shapes])
    w5, b5 = p[off[8]:off[9]].reshape(A, 512), p[off[9]:off[10]]
    obs8 = rng.integers(0, 256, (8, 4, 1, 84, 84), dtype=np.uint8)
    scale = np.abs(a.qvalues(obs8)).max()
    w5[0] *= 3.0; b5[0] = abs(b5[0]) + scale          # action 0 leads everywhere ...
    w5[1] = w5[0]; b5[1] = b5[0]                       # ... action 1 ties with it exactly ...
    w5[2] = w5[0]; b5[2] = b5[0] + np.float32(1e-4) * scale   # ... and action 2 is ahead by a clear margin
    a.set_params(p, "qnet")
    q8 = a.qvalues(obs8)                                                             # acting kernels
    q9 = a.qvalues(np.concatenate([obs8, rng.integers(0, 256, (1, 4, 1, 84, 84), dtype=np.uint8)]))[:8]   # training forward
    assert (q8[:, 0] == q8[:, 1]).all() and (q9[:, 0] == q9[:, 1]).all()             # the tie is exact on both paths
    assert (q8.argmax(1) == 2).all() and (q9.argmax(1) == 2).all()                   # the clear leader wins on both
    assert np.abs(q8 - q9).max() <= 5e-6 * np.abs(q9).max()
    b5[2] = b5[0] - 1.0                                                              # without the leader: the exact tie -> first maximum, on both
    a.set_params(p, "qnet")
    assert (a.sample_greedy(obs8) == 0).all()
    assert (a.sample_greedy(np.concatenate([obs8, obs8[:1]]))[:8] == 0).all()
    a.close()


def test_iqn_acting_kernels_agree_with_the_training_forward(B):
    """bdr_iqn_qvalues on <= 8 observations: the trunk's conv2 / conv3 and the merge layer ([33 n][3136] x [3136][512] at the Const32 percent
    points of iqn/model/base.rs:361-364) run on the acting kernels (act_small.hpp); more rows run the training forward.  The same rows
    through both: 1e-6 relative at most (same exact-f32 products, other order of additions)."""
    rng = np.random.default_rng(5)
    a = B.Iqn.build(B.IqnConfig(n_actions=9, device=0, batch_size=32, seed=3))
    a.eval()
    for n in (1, 2, 5, 8):
        obs = rng.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8)
        q_act = a.qvalues(obs)
        q_train = a.qvalues(np.concatenate([obs, rng.integers(0, 256, (12, 4, 1, 84, 84), dtype=np.uint8)]))[:n]
        assert q_act.shape == (n, 9)
        assert np.abs(q_act - q_train).max() <= 1e-6 * np.abs(q_train).max(), n
    a.close()
