"""Synchronous data-parallel mode (SURVEY.md 8(e) "Collective", last sentence): N ranks take the gradient of B/N rows each,
the gradient arenas are averaged (ncclAllReduce sum / N in production) and every rank takes the same optimizer step - which
must be the step ONE rank takes on the concatenated batch.  That statement is checkable on one GPU: two agents in one process,
halves of one minibatch, gradients averaged on the host, against the C oracle's full-batch step; plus the production path
(bdr_agent_set_grad_comm) over a 1-rank RCCL communicator, which must reproduce the fused single-GPU step bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def cnn(B, bs, **kw):
    return B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                                   device=0, batch_size=bs, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, **kw))


def mlp(B, bs, **kw):
    return B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2), opt_config=B.OptimizerConfig.Adam(1e-3)),
                                   device=0, batch_size=bs, critic_loss="Mse", tau=0.01, soft_update_interval=1, **kw))


def test_two_ranks_of_128_equal_one_rank_of_256_nature_cnn(B):
    from oracle import oracle as O
    from oracle import torch_ref as T
    from tests.test_gpu_dqn import assert_grads_close
    shapes = T.cnn_shapes(6)
    p0 = T.init_params(shapes, 7)
    ranks = [cnn(B, 128), cnn(B, 128)]
    full = cnn(B, 256)
    for a in ranks + [full]:
        a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.cnn_cfg(6), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000)
    # step 0 carries the parity claim (every side starts from the same parameters and Adam state): mean of the two half-batch
    # gradients == the oracle's full-batch gradient, and the step taken with it == the oracle's B = 256 step.  Steps 1, 2 check
    # that the ranks stay in lock step and track this library's own fused B = 256 agent (by then a ReLU-boundary flip of
    # step 0 may have moved individual channels by up to 2 lr on either side, so the oracle is no longer the same function).
    for step in range(3):
        obs, act, nobs, rew, term = T.synthetic_atari_batch(256, 6, 500 + step)
        halves = [tuple(x[r * 128:(r + 1) * 128] for x in (obs, act, nobs, rew, term)) for r in range(2)]
        before = [a.get_params("qnet") for a in ranks]
        recs = [a.grads_on_batch(*h) for a, h in zip(ranks, halves)]
        assert all((a.get_params("qnet") == b).all() for a, b in zip(ranks, before)) and all(a.n_opts == step for a in ranks)   # backward moves nothing
        g = 0.5 * (ranks[0].get_params("grad").astype(np.float64) + ranks[1].get_params("grad").astype(np.float64))
        if step == 0:
            r = ref.update(obs, act, nobs, rew, term, probe=True)
            assert_grads_close(g.astype(np.float32), r["grads"], shapes)                              # mean gradient == full-batch gradient
            assert abs(0.5 * (recs[0]["loss"] + recs[1]["loss"]) - r["loss"]) <= 1e-4 * abs(r["loss"])
        for a in ranks:
            a.set_params(g.astype(np.float32), "grad")
            a.apply_grads()
        full.update_on_batch(obs, act, nobs, rew, term)
        pa, pb = ranks[0].get_params("qnet"), ranks[1].get_params("qnet")
        assert (pa == pb).all() and ranks[0].n_opts == step + 1                                        # lock step, bit for bit
        if step == 0:   # vs the oracle's B=256 step: everything but the (<= 2 per layer) ReLU-flipped channels within 5 % of lr
            dp = np.abs(pa.astype(np.float64) - ref.q)
            assert (dp > 0.05 * 1e-4).sum() <= 2 * 257 and dp.max() <= 2.5e-4, ((dp > 0.05 * 1e-4).sum(), dp.max())
        dq = np.abs(pa.astype(np.float64) - full.get_params("qnet"))
        assert (dq > 0.1 * 1e-4).mean() < 2e-3 and dq.max() <= (step + 1) * 2.01e-4, (step, (dq > 0.1 * 1e-4).mean(), dq.max())
    for a in ranks + [full]:
        a.close()


def test_two_ranks_equal_one_rank_mlp_five_steps(B):
    from oracle import oracle as O
    from oracle import torch_ref as T
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 3)
    ranks = [mlp(B, 16), mlp(B, 16)]
    for a in ranks:
        a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.mlp_cfg(4, [64, 64], 2), p0, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1)
    rng = np.random.default_rng(0)
    for step in range(5):
        obs, nobs = rng.standard_normal((32, 4)).astype(np.float32), rng.standard_normal((32, 4)).astype(np.float32)
        act, rew, term = rng.integers(0, 2, 32), rng.standard_normal(32).astype(np.float32), (rng.random(32) < .1).astype(np.int8)
        for r_, a in enumerate(ranks):
            a.grads_on_batch(obs[r_ * 16:(r_ + 1) * 16], act[r_ * 16:(r_ + 1) * 16], nobs[r_ * 16:(r_ + 1) * 16], rew[r_ * 16:(r_ + 1) * 16], term[r_ * 16:(r_ + 1) * 16])
        g = (0.5 * (ranks[0].get_params("grad").astype(np.float64) + ranks[1].get_params("grad"))).astype(np.float32)
        r = ref.update(obs, act, nobs, rew, term, probe=True)
        assert rel(g, r["grads"]) < 2e-5, (step, rel(g, r["grads"]))
        for a in ranks:
            a.set_params(g, "grad"); a.apply_grads()
        assert (ranks[0].get_params("qnet") == ranks[1].get_params("qnet")).all()
        assert rel(ranks[0].get_params("qnet"), ref.q) < 1e-5, (step, rel(ranks[0].get_params("qnet"), ref.q))
        assert rel(ranks[0].get_params("qnet_tgt"), ref.q_tgt) < 1e-5                                 # soft update every opt, after the step
    for a in ranks:
        a.close()


@pytest.mark.parametrize("kind", ["cnn", "mlp"])
def test_grad_comm_over_one_rank_rccl_reproduces_the_fused_step(B, kind):
    """Production path: bdr_agent_set_grad_comm -> Agent::opt = backward, ncclAllReduce(grad) / nranks, optimizer step.  With a
    1-rank communicator the all-reduce is the identity, and the split step must give the fused step's parameters bit for bit
    (same element formulas, k_reduce_adam / k_adam un-fused), over the replay ring, including target syncs."""
    L = B._lib.lib()
    uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
    B._lib.check(L.bdr_comm_get_unique_id(uid))
    h = C.c_void_p()
    B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, 0, C.byref(h)))
    outs = []
    for use_comm in (False, True):
        if kind == "cnn":
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=600, seed=42), (4, 1, 84, 84), np.uint8)
            rb.fill_synthetic(600, seed=3, kind=0, n_actions=6)
            a = B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                                        device=0, batch_size=32, critic_loss="SmoothL1", tau=0.5, soft_update_interval=4, param_seed=9))
        else:
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=600, seed=42), (4,), np.float32)
            rb.fill_synthetic(600, seed=3, kind=1, n_actions=2)
            a = mlp(B, 32, param_seed=9)
        if use_comm:
            B._lib.check(L.bdr_agent_set_grad_comm(a.handle, h))
        for _ in range(9):
            a.opt(rb)
        rec = a.opt_with_record(rb)
        outs.append((a.get_params("qnet"), a.get_params("qnet_tgt"), a.get_params("exp_avg_sq"), rec["loss"], a.n_opts))
        if use_comm:
            B._lib.check(L.bdr_agent_set_grad_comm(a.handle, None))
            a.opt(rb); a.sync()
        a.close(); rb.close()
    (p0, t0, v0, l0, n0), (p1, t1, v1, l1, n1) = outs
    assert n0 == n1 == 10 and l0 == l1
    assert (p0 == p1).all() and (t0 == t1).all() and (v0 == v1).all()
    B._lib.check(L.bdr_comm_destroy(h))


def test_overlapped_parameter_exchange_is_the_identity_at_one_rank(B, monkeypatch):
    """bdr_agent_allreduce_params on the Nature-CNN agent runs per segment on the agent's communication queue, ordered against
    the two compute queues with device flags: the l1 / l2 segment as soon as its Adam pass is done (beside the conv dX tail),
    the conv segment behind k_reduce_adam; the next update's conv1 / l1 forward wait per segment.  With a 1-rank communicator
    the exchange must not change a single bit - every third step, every step, across soft updates, with host reads in between
    - and must equal the in-stream form (BDR_NO_XCHG_OVERLAP=1)."""
    L = B._lib.lib()
    uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
    B._lib.check(L.bdr_comm_get_unique_id(uid))
    h = C.c_void_p()
    B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, 0, C.byref(h)))

    def run(every, overlap, reads=False):
        if overlap:
            monkeypatch.delenv("BDR_NO_XCHG_OVERLAP", raising=False)
        else:
            monkeypatch.setenv("BDR_NO_XCHG_OVERLAP", "1")
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=800, seed=42), (4, 1, 84, 84), np.uint8)
        rb.fill_synthetic(800, seed=3, kind=0, n_actions=6)
        a = B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                                    device=0, batch_size=32, critic_loss="SmoothL1", tau=0.5, soft_update_interval=4, param_seed=9))
        mid = None
        for s in range(1, 26):
            a.opt(rb)
            if every and s % every == 0:
                B._lib.check(L.bdr_agent_allreduce_params(a.handle, h, 0))
            if reads and s == 13:
                mid = a.get_params("qnet")          # a host read while an exchange may be pending
        rec = a.opt_with_record(rb)
        out = (a.get_params("qnet"), a.get_params("qnet_tgt"), a.get_params("exp_avg"), rec["loss"], mid)
        a.close(); rb.close()
        return out
    base = run(0, True, reads=True)
    for every, overlap, reads in ((3, True, False), (1, True, True), (4, True, False), (3, False, False)):
        got = run(every, overlap, reads)
        assert (got[0] == base[0]).all() and (got[1] == base[1]).all() and (got[2] == base[2]).all() and got[3] == base[3], (every, overlap)
        if reads:
            assert (got[4] == base[4]).all()
    B._lib.check(L.bdr_comm_destroy(h))


def test_parameter_exchange_entry_points_cover_every_agent_kind(B):
    """bdr_agent_allreduce_params / bdr_agent_broadcast_params on the models bench.py exchanges for each configuration (DQN Mlp
    `qnet`, IQN `iqn`, SAC `pi` - SyncModel ships only the actor, sac/base.rs:377-386): over a 1-rank communicator both are the
    identity, bit for bit, and the agent keeps training afterwards."""
    L = B._lib.lib()
    uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
    B._lib.check(L.bdr_comm_get_unique_id(uid))
    h = C.c_void_p()
    B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, 0, C.byref(h)))
    rng = np.random.default_rng(0)
    agents = [
        (mlp(B, 32, param_seed=3), "qnet", lambda a: a.update_on_batch(rng.standard_normal((32, 4)).astype(np.float32), rng.integers(0, 2, 32),
                                                                        rng.standard_normal((32, 4)).astype(np.float32), np.ones(32, np.float32), np.zeros(32, np.int8))),
        (B.Iqn.build(B.IqnConfig(f_config=B.MlpConfig(in_dim=4, units=(32,), out_dim=16), feature_dim=16, embed_dim=8, m_units=(32,), n_actions=3,
                                 lr=1e-3, batch_size=8, device=0, seed=1)), "iqn", None),
        (B.Sac.build(B.SacConfig(obs_dim=5, act_dim=2, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=16, device=0, seed=2)), "pi", None),
    ]
    for a, which, step in agents:
        if a is None:
            continue
        before = a.get_params(which).copy()
        B._lib.check(L.bdr_agent_allreduce_params(a.handle, h, a.WHICH[which]))
        B._lib.check(L.bdr_agent_broadcast_params(a.handle, h, a.WHICH[which], 0))
        a.sync()
        assert (a.get_params(which) == before).all(), which
        if step is not None:
            step(a)
            assert not (a.get_params(which) == before).all()
        a.close()
    B._lib.check(L.bdr_comm_destroy(h))
