"""CPU checks of the exploration restatement (oracle/oracle.py::Explorer).

The reference has no tests for its explorers and draws from an unseeded global generator, so what can be
pinned is the arithmetic it specifies: the eps schedule (dqn/explorer.rs:69-70), the call structure (one
coin per call), argmax tie-breaking, and the uniformity / range of the draws."""
import numpy as np

from oracle.oracle import Explorer


def test_eps_schedule_matches_reference_formula():
    e = Explorer("eps_greedy", eps_start=1.0, eps_final=0.02, final_step=100_000)
    assert e.eps() == 1.0
    e.n_opts = 50_000
    assert abs(e.eps() - 0.51) < 1e-12
    e.n_opts = 100_000
    assert abs(e.eps() - 0.02) < 1e-12
    e.n_opts = 10_000_000
    assert e.eps() == 0.02          # clamped: .max(eps_final)


def test_eps_greedy_one_coin_per_call_and_counter():
    e = Explorer("eps_greedy", final_step=10, seed=3)
    q = np.array([[0.1, 0.9, 0.3], [0.5, 0.2, 0.5], [0.0, 0.0, 0.0]], np.float32)
    n_rand = 0
    for call in range(200):
        act, eps, is_random = e.sample(q, train=True)
        assert e.n_opts == call + 1
        assert act.shape == (3,) and act.min() >= 0 and act.max() < 3
        if not is_random:
            assert act.tolist() == [1, 0, 0]      # first maximum on ties, like Tensor::argmax
        n_rand += is_random
    # eps is 0.02 after 10 calls: about 10*~0.5 + 190*0.02 random calls
    assert 2 <= n_rand <= 25
    assert e.n_samples_act == 200


def test_draw_ranges_and_uniformity():
    e = Explorer(seed=11)
    f = np.array([e.f64() for _ in range(4000)])
    assert f.min() >= 0.0 and f.max() < 1.0 and abs(f.mean() - 0.5) < 0.02
    g = np.array([float(e.f32()) for _ in range(4000)])
    assert g.min() >= 0.0 and g.max() < 1.0 and abs(g.mean() - 0.5) < 0.02
    counts = np.bincount([e.below(6) for _ in range(12000)], minlength=6)
    assert counts.sum() == 12000 and counts.min() > 1800 and counts.max() < 2200


def test_softmax_frequencies_follow_softmax_probabilities():
    e = Explorer("softmax", seed=5)
    q = np.array([[1.0, 2.0, 0.5, -1.0]], np.float32)
    p = np.exp(q[0] - q[0].max()); p /= p.sum()
    acts = np.array([e.sample(q, train=True)[0][0] for _ in range(8000)])
    freq = np.bincount(acts, minlength=4) / 8000.0
    assert np.abs(freq - p).max() < 0.02


def test_eval_is_greedy_with_rare_random_action_for_dqn_only():
    q = np.array([[0.0, 3.0, 1.0]] * 2, np.float32)
    e = Explorer(seed=9)
    n_rand = sum(e.sample(q, train=False, dqn=True)[2] for _ in range(5000))
    assert 20 <= n_rand <= 90                     # 1 %
    e2 = Explorer(seed=9)
    for _ in range(100):
        act, _, is_random = e2.sample(q, train=False, dqn=False)
        assert not is_random and act.tolist() == [1, 1]
