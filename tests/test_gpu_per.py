"""Prioritized replay on the device vs the CPU restatement (oracle/oracle.py::{SumTree, PerReplay}) and the
reference's own known answers (sum_tree.rs:180-217).  Row (f)-2 of the scope table.

Index work is held to exact equality.  With alpha == 1 the transformed priorities are exact in any powf, so
the whole f32 tree must match bit for bit; with alpha == 0.6 the leaves go through powf (glibc on the CPU,
double pow rounded to f32 on the device) and are held to 1 ulp."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DATA = [0.5, 0.2, 0.8, 0.3, 1.1, 2.5, 3.9]


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _buf(B, cap, seed=42, **per):
    cfg = B.SimpleReplayBufferConfig(capacity=cap, seed=seed, per_config=B.PerConfig(**per))
    return B.SimpleReplayBuffer(cfg, (4,), np.float32)


def _push(rb, n, rng):
    rb.push(rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)), rng.standard_normal((n, 4)).astype(np.float32),
            np.ones(n, np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))


def test_reference_sum_tree_kats_on_device(B):
    """sum_tree.rs:184-199: capacity 8, alpha 1, priorities DATA -> get() table."""
    rb = _buf(B, 8, alpha=1.0, normalize="Batch")
    _push(rb, 7, np.random.default_rng(0))
    rb.update_priority(np.arange(7), np.array(DATA, np.float32))
    for s, want in [(0.0, 0), (0.4, 0), (0.5, 0), (0.6, 1), (1.2, 2), (1.6, 3), (2.0, 4), (2.8, 4)]:
        assert rb.per_get(s) == want, (s, rb.per_get(s), want)
    info = rb.per_info()
    assert info["n_samples"] == 7 and info["n_opts"] == 1
    assert abs(info["total"] - sum(DATA)) < 1e-5
    rb.close()


@pytest.mark.parametrize("cap,alpha,norm", [(8, 1.0, "Batch"), (100, 1.0, "All"), (1000, 0.6, "All"), (37, 0.6, "Batch")])
def test_tree_indices_and_weights_match_oracle(B, cap, alpha, norm):
    """push / batch / update_priority sequences replayed on the CPU restatement: same tree, same indices,
    same weights (non-power-of-two capacities included: the tree keeps the reference's array layout)."""
    from oracle.oracle import PerReplay
    rng = np.random.default_rng(cap)
    rb = _buf(B, cap, seed=7, alpha=alpha, normalize=norm, n_opts_final=40)
    ref = PerReplay(cap, 7, alpha=alpha, normalize=norm, n_opts_final=40)
    exact = alpha == 1.0
    for rnd in range(30):
        n_push = int(rng.integers(1, max(2, cap // 3)))
        _push(rb, n_push, rng); ref.push(n_push)
        for _ in range(2):
            n = int(rng.integers(1, 65))
            b = rb.batch(n)
            ixs, ws = ref.batch(n)
            t_dev, t_ref = rb.per_tree(), ref.tree.tree()
            if exact:
                assert (t_dev == t_ref).all(), rnd
            else:
                np.testing.assert_allclose(t_dev, t_ref, rtol=3e-7, atol=0)
            assert b.ix_sample.tolist() == ixs.tolist(), (rnd, n)
            np.testing.assert_allclose(b.weight, ws, rtol=5e-6)
            # duplicates inside one update batch chain like the sequential loop of base.rs:421-423
            td = (rng.random(n).astype(np.float32) * 3.0).astype(np.float32)
            rb.update_priority(b.ix_sample, td); ref.update_priority(ixs, td)
        info = rb.per_info()
        assert info["n_samples"] == ref.tree.n_samples and info["n_opts"] == ref.n_opts
        assert abs(info["beta"] - ref.beta()) < 1e-7
    t_dev, t_ref = rb.per_tree(), ref.tree.tree()
    if exact:
        assert (t_dev == t_ref).all()
    else:
        np.testing.assert_allclose(t_dev, t_ref, rtol=3e-7, atol=0)
    rb.close()


def test_batch_rows_follow_the_sampled_indices(B):
    rng = np.random.default_rng(3)
    rb = _buf(B, 64, alpha=0.6)
    n = 50
    obs = rng.standard_normal((n, 4)).astype(np.float32)
    rb.push(obs, np.arange(n).reshape(n, 1), obs + 1, np.arange(n, dtype=np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))
    b = rb.batch(32)
    ix = b.ix_sample.astype(np.int64)
    assert (b.obs == obs[ix]).all() and (b.act.ravel() == ix).all() and (b.reward == ix.astype(np.float32)).all()
    assert b.weight is not None and b.weight.shape == (32,) and (b.weight > 0).all() and b.weight.max() <= 1.0 + 1e-6
    rb.close()


def test_uniform_buffer_has_no_weights_and_update_priority_is_a_noop(B):
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=16, seed=1), (4,), np.float32)
    _push(rb, 8, np.random.default_rng(0))
    b = rb.batch(4)
    assert b.weight is None
    rb.update_priority(b.ix_sample, np.ones(4, np.float32))      # base.rs:414: no per_state -> nothing happens
    with pytest.raises(B.BdrError):
        rb.per_info()
    rb.close()


def _cart_per(s):
    rng = np.random.default_rng(300 + s)
    obs = rng.standard_normal((32, 4)).astype(np.float32)
    nobs = rng.standard_normal((32, 4)).astype(np.float32)
    act = rng.integers(0, 2, 32)
    term = (rng.random(32) < 0.1).astype(np.int8)
    rew = (np.ones(32, np.float32) * np.random.default_rng(700 + s).uniform(-2, 2, 32)).astype(np.float32)
    return obs, act, nobs, rew, term


@pytest.mark.parametrize("name,kw", [("dqn_mlp_per_huber", dict(critic_loss="SmoothL1", param_seed=4)),
                                     ("dqn_mlp_per_mse_clip", dict(critic_loss="Mse", clip_td_err=(0.05, 0.9), double_dqn=True, param_seed=5))])
def test_weighted_update_critic_matches_goldens(B, golden_dir, name, kw):
    """The `if let Some(ws) = weight` branch of update_critic (dqn/base.rs:123-145) vs the committed ATen fixtures."""
    from oracle import torch_ref as T
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden import per_weights
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    shapes = T.mlp_shapes(4, [64, 64], 2)
    seed = kw.pop("param_seed")
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2),
                                                    opt_config=B.OptimizerConfig.Adam(1e-3)),
                      device=0, batch_size=32, tau=0.01, soft_update_interval=1, **kw)
    a = B.Dqn.build(cfg)
    p0 = T.init_params(shapes, seed)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    for s in range(3):
        rec = a.update_on_batch(*_cart_per(s), weight=per_weights(s, 32))
        np.testing.assert_allclose(rec["td_errs"], g[f"s{s}_td_errs"], rtol=1e-4, atol=1e-6)
        assert abs(rec["loss"] - g[f"s{s}_loss"]) <= 1e-4 * abs(g[f"s{s}_loss"]) + 1e-8, s
        gr, ref = a.get_params("grad").astype(np.float64), g[f"s{s}_grads_sample"].astype(np.float64)
        assert np.abs(gr - ref).max() <= 2e-4 * np.abs(ref).max(), s
        d = np.abs(a.get_params("qnet").astype(np.float64) - g[f"s{s}_params_sample"])
        assert d.max() < 0.05 * 1e-3, (s, d.max())
    a.close()


def test_dqn_opt_over_per_buffer_updates_priorities(B):
    """Agent::opt on a PER buffer == oracle pipeline: sample via the tree, weighted loss, update_priority(td)."""
    from oracle.oracle import PerReplay
    from oracle import torch_ref as T
    rng = np.random.default_rng(9)
    cap, Bsz, n = 256, 32, 200
    rb = _buf(B, cap, seed=42, alpha=1.0, normalize="All", n_opts_final=50)
    ref = PerReplay(cap, 42, alpha=1.0, normalize="All", n_opts_final=50)
    obs = rng.standard_normal((n, 4)).astype(np.float32)
    nobs = rng.standard_normal((n, 4)).astype(np.float32)
    act = rng.integers(0, 2, (n, 1)).astype(np.int64)
    rew = rng.uniform(-2, 2, n).astype(np.float32)
    term = (rng.random(n) < 0.1).astype(np.int8)
    rb.push(obs, act, nobs, rew, term, np.zeros(n, np.int8)); ref.push(n)
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 13)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2),
                                                    opt_config=B.OptimizerConfig.Adam(1e-3)),
                      device=0, batch_size=Bsz, tau=0.01, soft_update_interval=1, critic_loss="SmoothL1")
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn("mlp", shapes, p0, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1)
    for step in range(6):
        ixs, ws = ref.batch(Bsz)
        r = t.update(obs[ixs], act[ixs, 0], nobs[ixs], rew[ixs], term[ixs], weight=ws)
        ref.update_priority(ixs, r["td_abs"])
        rec = a.opt_with_record(rb)
        assert abs(rec["loss"] - r["loss"]) <= 2e-4 * abs(r["loss"]) + 1e-7, step
        # the device tree was updated with the device's own td errors (f32 round-off apart from the oracle's)
        np.testing.assert_allclose(rb.per_tree(), ref.tree.tree(), rtol=2e-4, atol=1e-6)
        assert rb.per_info()["n_opts"] == step + 1
    a.close(); rb.close()


def test_large_updates_and_wrapping_pushes_chunk_correctly(B):
    """update_priority with more rows than one device chunk (1024) and pushes that wrap / exceed the capacity:
    still the sequential semantics of base.rs:227-235 and :421-423."""
    from oracle.oracle import PerReplay
    rng = np.random.default_rng(21)
    cap = 1500
    rb = _buf(B, cap, seed=3, alpha=1.0, normalize="All")
    ref = PerReplay(cap, 3, alpha=1.0, normalize="All")
    for n_push in (1200, 700, 1, 1499):                 # second push wraps the ring
        _push(rb, n_push, rng); ref.push(n_push)
        assert (rb.per_tree() == ref.tree.tree()).all(), n_push
    assert rb.per_info()["n_samples"] == cap == ref.tree.n_samples
    ixs = rng.integers(0, cap, 3000).astype(np.uint64)   # many duplicates, 3 device chunks
    td = (rng.random(3000) * 2).astype(np.float32)
    rb.update_priority(ixs, td); ref.update_priority(ixs, td)
    assert (rb.per_tree() == ref.tree.tree()).all()
    b = rb.batch(1024)                                   # the largest PER batch
    r_ix, r_w = ref.batch(1024)
    assert b.ix_sample.tolist() == r_ix.tolist()
    np.testing.assert_allclose(b.weight, r_w, rtol=5e-6)
    with pytest.raises(B.BdrError):
        rb.batch(1025)
    rb.close()


def test_per_must_be_enabled_on_an_empty_buffer(B):
    import ctypes as C
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=16, seed=1), (4,), np.float32)
    _push(rb, 3, np.random.default_rng(0))
    rc = B._lib.lib().bdr_replay_enable_per(rb.handle, C.byref(B.PerConfig().to_c()))
    assert rc != 0
    rb.close()


@pytest.mark.parametrize("loss,clip,ddqn", [("SmoothL1", None, False), ("Mse", (0.05, 0.9), True)])
def test_weighted_update_critic_nature_cnn_vs_aten(B, loss, clip, ddqn):
    """The importance-weighted branch on the Nature-CNN agent (TD step fused into the head kernel) against the
    ATen restatement, B = 6, three steps."""
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(6)
    p0 = T.init_params(shapes, 17)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6),
                                                    opt_config=B.OptimizerConfig.Adam(1e-4)),
                      device=0, batch_size=6, tau=1.0, soft_update_interval=2, critic_loss=loss, clip_td_err=clip, double_dqn=ddqn)
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn("cnn", shapes, p0, lr=1e-4, critic_loss=loss, clip_td_err=clip, double_dqn=ddqn, tau=1.0, soft_update_interval=2)
    rng = np.random.default_rng(77)
    for s in range(3):
        obs, act, nobs, rew, term = T.synthetic_atari_batch(6, 6, 400 + s)
        rew = (rew + rng.uniform(-1.5, 1.5, 6)).astype(np.float32)       # |td| on both sides of the Huber knee / clip range
        w = (0.2 + 0.8 * rng.random(6)).astype(np.float32)
        r = t.update(obs, act, nobs, rew, term, weight=w)
        rec = a.update_on_batch(obs, act, nobs, rew, term, weight=w)
        np.testing.assert_allclose(rec["td_errs"], r["td_abs"], rtol=2e-4, atol=1e-6)
        assert abs(rec["loss"] - r["loss"]) <= 2e-4 * abs(r["loss"]) + 1e-8, s
        g, gr = a.get_params("grad").astype(np.float64), r["grads"].astype(np.float64)
        assert np.abs(g - gr).max() <= 5e-4 * np.abs(gr).max(), s
        assert np.abs(a.get_params("qnet").astype(np.float64) - t.params()).max() < 0.05 * 1e-4
    a.close()


def test_per_opt_stream_is_deterministic_and_overlap_invariant(B):
    """100 opt steps of the Nature-CNN agent over a PER ring: parameters and the priority tree are bit-identical
    between two runs and between the three-queue schedules (TD / weight gradients / tree update; BDR_SCHED 1 = ordered by
    events, 3 = default, ordered by device flags) and the serial one (0)."""
    def run(sched):
        if sched is None:
            os.environ.pop("BDR_SCHED", None)
        else:
            os.environ["BDR_SCHED"] = str(sched)
        try:
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=5_000, seed=42, per_config=B.PerConfig(n_opts_final=60)),
                                      (4, 1, 84, 84), "uint8")
            rb.fill_synthetic(5_000, seed=3, kind=0, n_actions=6)
            cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6),
                                                            opt_config=B.OptimizerConfig.Adam(1e-4)),
                              device=0, batch_size=32, critic_loss="SmoothL1", tau=1.0, soft_update_interval=40, param_seed=5)
            a = B.Dqn.build(cfg)
            a.train()
            for _ in range(100):
                a.opt(rb)
            a.sync()
            out = a.get_params("qnet"), rb.per_tree(), rb.per_info()["n_opts"]
            a.close(); rb.close()
            return out
        finally:
            os.environ.pop("BDR_SCHED", None)

    p1, t1, n1 = run(None)
    assert n1 == 100 and np.isfinite(p1).all() and np.isfinite(t1).all()
    for sched in (None, 0, 1, 3):
        p, t, n = run(sched)
        assert n == 100 and (p1 == p).all() and (t1 == t).all(), sched


def test_nan_priority_is_reported_and_the_tree_stays_finite(B):
    """SumTree::update panics when `change` is NaN (sum_tree.rs:101-104).  The library never aborts: the offending updates
    are dropped (leaf and partial sums keep their values), a device flag is raised and the call reports BDR_ERR_INVALID."""
    cap = 64
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=1, per_config=B.PerConfig(alpha=1.0)), (4,), np.float32)
    rng = np.random.default_rng(0)
    n = 40
    rb.push(rng.standard_normal((n, 4)).astype(np.float32), np.zeros((n, 1), np.int64), rng.standard_normal((n, 4)).astype(np.float32),
            np.zeros(n, np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))
    rb.update_priority(np.arange(8), np.linspace(0.1, 0.8, 8).astype(np.float32))
    before = rb.per_tree()
    with pytest.raises(B.BdrError) as e:
        rb.update_priority(np.array([3, 5, 7]), np.array([0.5, np.nan, 0.25], np.float32))
    assert e.value.code == 1 and "NaN" in str(e.value)
    after = rb.per_tree()
    assert np.isfinite(after).all()
    leaves = cap - 1
    assert after[leaves + 5] == before[leaves + 5]                      # the NaN update was dropped
    assert after[leaves + 3] != before[leaves + 3] and after[leaves + 7] != before[leaves + 7]   # the others were applied
    rb.update_priority(np.array([1]), np.array([0.3], np.float32))       # the flag was cleared
    assert np.isfinite(rb.per_info()["total"])
    # a rejected synthetic fill must not have touched the ring (the precondition is checked first)
    o0 = rb.read_rows(0, 4)[0].copy()
    with pytest.raises(B.BdrError):
        rb.fill_synthetic(10, seed=1, kind=1, n_actions=2)
    assert (rb.read_rows(0, 4)[0] == o0).all()
    rb.close()
