"""HIP DQN step (through the C ABI) vs the CPU oracle and the committed PyTorch goldens.

Bar (BASELINE.json north_star): Q-values within 1e-4 relative for a fixed seed and fixed minibatch.
Gradients / parameters are held to the same class of tolerance (f32 summation-order noise)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QTOL = 1e-4  # relative, on Q-values (north_star)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def assert_grads_close(g, ref, shapes, tol=2e-4, flip_tol=5e-2, max_flipped=2):
    """Per-variable gradient check that tolerates ReLU-boundary flips: a unit whose pre-activation is
    within f32 round-off of zero may be masked differently by two correct implementations, which
    perturbs exactly ONE output channel of that layer's weight/bias gradient (seen at B=256 on
    conv1: 102400x32 units).  Every other entry must agree to `tol` (relative to the variable's max)."""
    o = 0
    for sh in shapes:
        n = int(np.prod(sh))
        a, b = g[o:o + n].astype(np.float64), ref[o:o + n].astype(np.float64)
        d = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
        bad = (d > tol).reshape(sh[0], -1).any(1)
        assert bad.sum() <= max_flipped and d.max() < flip_tol, (sh, int(bad.sum()), d.max())
        o += n


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def make_agent(B, A=6, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=A),
                                                    opt_config=B.OptimizerConfig.Adam(kw.pop("lr", 1e-4))),
                      device=0, **kw)
    return B.Dqn.build(cfg)


def test_param_roundtrip_and_layout(B):
    from oracle import torch_ref as T
    a = make_agent(B, batch_size=4)
    p = T.init_params(T.cnn_shapes(6), 11)
    a.set_params(p, "qnet")
    assert (a.get_params("qnet") == p).all()
    assert a.param_count() == 1687206
    a.close()


def test_qvalues_match_oracle(B):
    from oracle import oracle as O
    from oracle import torch_ref as T
    a = make_agent(B, batch_size=8)
    p = T.init_params(T.cnn_shapes(6), 5)
    a.set_params(p, "qnet")
    obs = np.random.default_rng(0).integers(0, 256, (13, 4, 1, 84, 84), dtype=np.uint8)  # ragged M tails
    q = a.qvalues(obs)
    qo = O.net_forward(O.cnn_cfg(6), p, obs)
    assert rel(q, qo) < QTOL, rel(q, qo)
    assert (a.sample_greedy(obs) == qo.argmax(-1)).all()
    a.close()


def _run_golden(B, fix, batch_fn, n_steps, param_seed, Bsz, **kw):
    from oracle import torch_ref as T
    g = np.load(fix)
    shapes = T.cnn_shapes(6)
    a = make_agent(B, batch_size=Bsz, **kw)
    p0 = T.init_params(shapes, param_seed)
    a.set_params(p0, "qnet")
    a.set_params(p0, "qnet_tgt")
    lr = kw.get("lr", 1e-4)
    for s in range(n_steps):
        obs, act, nobs, rew, term = batch_fn(s)
        rec = a.update_on_batch(obs, act, nobs, rew, term)
        assert rel(a.probe("q_pred_all", Bsz * 6), g[f"s{s}_q_pred_all"].ravel()) < QTOL, s
        assert rel(a.probe("q_next_all", Bsz * 6), g[f"s{s}_q_next_all"].ravel()) < QTOL, s
        assert rel(a.probe("tgt", Bsz), g[f"s{s}_tgt"]) < QTOL
        assert abs(rec["loss"] - g[f"s{s}_loss"]) <= QTOL * abs(g[f"s{s}_loss"]) + 1e-9
        grads = a.get_params("grad")
        st = max(1, grads.size // 4096) | 1
        # every sampled gradient entry to 2e-4 of the largest; ONE entry may sit at a ReLU boundary (a unit whose pre-activation is within f32
        # round-off of zero is masked differently by two correct implementations; measured with tools/diag/golden_b8_modes.py: step 0 3e-7 /
        # 8e-8, step 1 2.8e-7 with exact products and ONE conv3 entry at 2.4e-4 with the split forward, all others below 1e-4) - bounded at 1e-3
        dg = np.abs(grads[::st].astype(np.float64) - g[f"s{s}_grads_sample"]) / np.abs(g[f"s{s}_grads_sample"]).max()
        assert (dg > 2e-4).sum() <= 1 and dg.max() < 1e-3, (s, int((dg > 2e-4).sum()), dg.max())
        o = 0
        for i, sh in enumerate(shapes):
            n = int(np.prod(sh))
            gn = np.linalg.norm(grads[o:o + n].astype(np.float64))
            assert abs(gn - g[f"s{s}_grad_norms"][i]) <= 1e-3 * g[f"s{s}_grad_norms"][i] + 1e-12, (s, i)
            o += n
        d = np.abs(a.get_params("qnet")[::st].astype(np.float64) - g[f"s{s}_params_sample"])
        assert d.max() < 0.05 * lr, (s, d.max())
        assert rel(a.get_params("qnet_tgt")[::st], g[f"s{s}_tgt_params_sample"]) < 1e-5
    a.close()


def test_golden_cnn_b4_huber(B, golden_dir):
    from oracle import torch_ref as T
    _run_golden(B, os.path.join(golden_dir, "dqn_cnn_b4_huber.npz"), lambda s: T.synthetic_atari_batch(4, 6, 100 + s), 3, 1, 4,
                lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=2)


def test_golden_cnn_b8_mse_double_dqn(B, golden_dir):
    from oracle import torch_ref as T
    _run_golden(B, os.path.join(golden_dir, "dqn_cnn_b8_mse_ddqn.npz"), lambda s: T.synthetic_atari_batch(8, 6, 200 + s), 2, 2, 8,
                lr=1e-4, critic_loss="Mse", double_dqn=True, tau=0.005, soft_update_interval=1)


# Atari's minimal action sets go from 3 to 18 actions (border-atari-env/src/env.rs:97-103); k_head / k_head_bwd are compiled for
# register blocks of 8, 24 and 64 actions (csrc/dqn.hip forward()): 4 and 6 take <.., 8>, 9 and 18 <.., 24>, 33 <.., 64>
@pytest.mark.parametrize("A", [6, 4, 9, 18, 33])
def test_full_batch_256_vs_oracle(B, A):
    """BASELINE config: B=256, SmoothL1 (A=6 is the headline shape).  One step against the C oracle (fixed minibatch)."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(A)
    p0 = T.init_params(shapes, 7)
    obs, act, nobs, rew, term = T.synthetic_atari_batch(256, A, 77)
    assert act.max() == A - 1 and act.min() == 0   # every column of the head's register block is selected by some row
    term[:8] = 1  # make sure the (1 - is_terminated) branch is exercised
    a = make_agent(B, A=A, batch_size=256, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.cnn_cfg(A), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000)
    rec = a.update_on_batch(obs, act, nobs, rew, term)
    r = ref.update(obs, act, nobs, rew, term, probe=True)
    assert rel(a.probe("q_pred_all", 256 * A), r["q_pred_all"].ravel()) < QTOL
    assert rel(a.probe("q_next_all", 256 * A), r["q_next_all"].ravel()) < QTOL
    assert rel(a.probe("pred", 256), r["pred"]) < QTOL and rel(a.probe("tgt", 256), r["tgt"]) < QTOL
    assert abs(rec["loss"] - r["loss"]) <= QTOL * abs(r["loss"])
    assert_grads_close(a.get_params("grad"), r["grads"], shapes)
    # parameters after the Adam step: the first step moves every weight by ~lr*sign(g), so the (at most
    # max_flipped) ReLU-flipped channels may differ by up to 2*lr; everything else agrees to 5% of lr
    dp = np.abs(a.get_params("qnet").astype(np.float64) - ref.q)
    assert (dp > 0.05 * 1e-4).sum() <= 2 * 257 and dp.max() <= 2.5e-4
    a.close()


def test_opt_over_replay_matches_oracle_pipeline(B):
    """Agent::opt over the HBM ring == oracle replay + oracle update on the same transitions:
    sampled indices bit-identical, Q-values within tolerance, n_opts / soft-update bookkeeping."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    from tests import synth
    cap, Bsz = 512, 32
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4, 1, 84, 84), np.uint8)
    rb.fill_synthetic(cap, seed=3, kind=0, n_actions=6)
    rows = synth.atari_rows(3, 0, cap)
    oref = O.Replay(cap, 42, 28224, 8)
    oref.push(rows[0], rows[1].reshape(-1, 1), rows[2], rows[3], rows[4], rows[5])
    p0 = T.init_params(T.cnn_shapes(6), 9)
    a = make_agent(B, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=2, record_verbose_level=2)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.cnn_cfg(6), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=2)
    for step in range(3):
        # every step carries the parity bar (north_star: 1e-4 on Q-values for a fixed seed and minibatch): the oracle takes each step from the
        # state the device is in - parameters, target parameters and both Adam moments copied over before the step (rounds 1-5 let the two
        # sides drift and allowed 3e-4 from step 1 on: Adam's first steps move a weight whose gradient is within round-off of 0 by +-lr on
        # either side, which is a property of the optimizer, not of this path)
        if step > 0:
            ref.q[:], ref.q_tgt[:] = a.get_params("qnet"), a.get_params("qnet_tgt")
            ref.m[:], ref.v[:] = a.get_params("exp_avg"), a.get_params("exp_avg_sq")
        rec = a.opt_with_record(rb)
        b = oref.batch(Bsz)
        r = ref.update(b["obs"].reshape(Bsz, 4, 1, 84, 84), b["act"].view(np.int64).ravel(),
                       b["next_obs"].reshape(Bsz, 4, 1, 84, 84), b["reward"], b["is_terminated"], probe=True)
        assert rel(a.probe("q_pred_all", Bsz * 6), r["q_pred_all"].ravel()) < QTOL, (step, rel(a.probe("q_pred_all", Bsz * 6), r["q_pred_all"].ravel()))
        assert rel(a.probe("q_next_all", Bsz * 6), r["q_next_all"].ravel()) < QTOL, step
        assert abs(rec["loss"] - r["loss"]) <= QTOL * abs(r["loss"]) + 1e-7
        assert abs(rec["reward_mean"] - b["reward"].mean()) < 1e-6
    assert a.n_opts == 3
    assert rel(a.get_params("qnet_tgt"), ref.q_tgt) < 1e-3
    a.close(); rb.close()


def test_save_load_roundtrip(B, tmp_path):
    a = make_agent(B, batch_size=4, param_seed=3)
    p = a.get_params("qnet")
    files = a.save_params(str(tmp_path))
    assert all(os.path.exists(f) for f in files)
    b = make_agent(B, batch_size=4, param_seed=99)
    assert not (b.get_params("qnet") == p).all()
    b.load_params(str(tmp_path))
    assert (b.get_params("qnet") == p).all() and (b.get_params("qnet_tgt") == p).all()
    a.close(); b.close()


def test_missing_device_config_is_an_error(B):
    with pytest.raises(B.BdrError):
        B.Dqn.build(B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(out_dim=6))))


def test_rccl_comm_single_rank(B):
    """The native RCCL path end to end on one GPU: unique id, ncclCommInitRank(nranks=1), and the
    all-reduce / broadcast entry points (identity at nranks == 1)."""
    import ctypes as C
    L = B._lib.lib()
    uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
    B._lib.check(L.bdr_comm_get_unique_id(uid))
    h = C.c_void_p()
    B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, 0, C.byref(h)))
    a = make_agent(B, batch_size=4, param_seed=5)
    p = a.get_params("qnet")
    B._lib.check(L.bdr_agent_allreduce_params(a.handle, h, 0))
    B._lib.check(L.bdr_agent_broadcast_params(a.handle, h, 0, 0))
    a.sync()
    assert (a.get_params("qnet") == p).all()
    B._lib.check(L.bdr_comm_destroy(h))
    a.close()


# ------------------------------------------------------------------------------------------------
# Dqn<E, Mlp, R> -- BASELINE config 1 (CartPole-shaped: MLP[64,64], replay 10k, batch 32)
def make_mlp_agent(B, in_dim=4, units=(64, 64), A=2, **kw):
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=in_dim, units=tuple(units), out_dim=A),
                                                    opt_config=B.OptimizerConfig.Adam(kw.pop("lr", 1e-3))),
                      device=0, **kw)
    return B.Dqn.build(cfg)


def _cart(s):
    rng = np.random.default_rng(300 + s)
    obs = rng.standard_normal((32, 4)).astype(np.float32)
    nobs = rng.standard_normal((32, 4)).astype(np.float32)
    act = rng.integers(0, 2, 32)
    return obs, act, nobs, np.ones(32, np.float32), (rng.random(32) < 0.1).astype(np.int8)


def test_mlp_golden_cartpole(B, golden_dir):
    """5 opt steps of the CartPole-shaped DQN against the committed PyTorch goldens."""
    from oracle import torch_ref as T
    g = np.load(os.path.join(golden_dir, "dqn_mlp_cartpole.npz"))
    shapes = T.mlp_shapes(4, [64, 64], 2)
    a = make_mlp_agent(B, batch_size=32, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1)
    p0 = T.init_params(shapes, 3)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    assert (a.get_params("qnet") == p0).all() and a.param_count() == 4610
    for s in range(5):
        rec = a.update_on_batch(*_cart(s))
        assert rel(a.probe("q_pred_all", 64), g[f"s{s}_q_pred_all"].ravel()) < QTOL, s
        assert rel(a.probe("q_next_all", 64), g[f"s{s}_q_next_all"].ravel()) < QTOL, s
        assert rel(a.probe("tgt", 32), g[f"s{s}_tgt"]) < QTOL
        assert abs(rec["loss"] - g[f"s{s}_loss"]) <= QTOL * abs(g[f"s{s}_loss"]) + 1e-9
        assert_grads_close(a.get_params("grad"), g[f"s{s}_grads_sample"], shapes)   # stride 1: full vectors
        d = np.abs(a.get_params("qnet").astype(np.float64) - g[f"s{s}_params_sample"])
        assert d.max() < 0.05 * 1e-3, (s, d.max())
        assert rel(a.get_params("qnet_tgt"), g[f"s{s}_tgt_params_sample"]) < 1e-5
    a.close()


def test_mlp_opt_over_replay(B):
    """Agent::opt over the HBM ring for the Mlp agent == oracle replay + oracle update (CartPole sizes)."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    rng = np.random.default_rng(5)
    cap, Bsz = 10000, 32
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4,), np.float32)
    oref = O.Replay(cap, 42, 16, 8)
    n = 500
    tr = (rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
          rng.standard_normal((n, 4)).astype(np.float32), np.ones(n, np.float32), (rng.random(n) < .1).astype(np.int8),
          np.zeros(n, np.int8))
    rb.push(*tr); oref.push(*tr)
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 11)
    a = make_mlp_agent(B, batch_size=Bsz, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, double_dqn=True)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.mlp_cfg(4, [64, 64], 2), p0, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1,
                      double_dqn=True)
    for step in range(4):
        rec = a.opt_with_record(rb)
        b = oref.batch(Bsz)
        r = ref.update(b["obs"].view(np.float32).reshape(Bsz, 4), b["act"].view(np.int64).ravel(),
                       b["next_obs"].view(np.float32).reshape(Bsz, 4), b["reward"], b["is_terminated"], probe=True)
        assert rel(a.probe("q_pred_all", Bsz * 2), r["q_pred_all"].ravel()) < 3e-4, step
        assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-7
    assert rel(a.get_params("qnet_tgt"), ref.q_tgt) < 1e-3
    q = a.qvalues(tr[0][:7])
    assert rel(q, O.net_forward(O.mlp_cfg(4, [64, 64], 2), a.get_params("qnet"), tr[0][:7])) < QTOL
    a.close(); rb.close()


def test_mlp_opt_over_a_xoshiro_indexed_buffer(B):
    """A buffer built with index_rng = xoshiro256++ (the device-native generator, replay.hip k_xo_indices): the step kernel's own
    StdRng draw is not used, the buffer's gather names the rows, and the update on those rows equals the oracle's."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    rng = np.random.default_rng(6)
    cap, Bsz, n = 600, 32, 500
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42, index_rng="xoshiro256++"), (4,), np.float32)
    tr = (rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
          rng.standard_normal((n, 4)).astype(np.float32), rng.standard_normal(n).astype(np.float32), (rng.random(n) < .1).astype(np.int8),
          np.zeros(n, np.int8))
    rb.push(*tr)
    lanes = O.XoshiroLanes(42)
    p0 = T.init_params(T.mlp_shapes(4, [64, 64], 2), 11)
    a = make_mlp_agent(B, batch_size=Bsz, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, double_dqn=True)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.mlp_cfg(4, [64, 64], 2), p0, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, double_dqn=True)
    for step in range(4):
        rec = a.opt_with_record(rb)
        ix = lanes.sample_indices(n, Bsz).astype(np.int64)
        r = ref.update(tr[0][ix], tr[1][ix, 0], tr[2][ix], tr[3][ix], tr[4][ix], probe=True)
        assert rel(a.probe("q_pred_all", Bsz * 2), r["q_pred_all"].ravel()) < 3e-4, step
        assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-7
    assert (rb.sample_indices(40) == lanes.sample_indices(n, 40)).all()      # four batches were drawn on both sides
    a.close(); rb.close()


def test_mlp_step_kernel_draws_its_own_batch(B, monkeypatch):
    """For nets that fit one workgroup the step kernel is also the replay buffer's sample (replay_sample_plan + the gather phase
    of k_dqn_mlp_step): same StdRng stream position, same rows as the separate gather launch - parameters after 12 opts with
    pushes in between are bit-identical, and both buffers continue their index stream from the same position.  The same holds
    for the LDS-resident variant of the kernel (k_dqn_mlp_step_lds) against the one that exchanges matrices through global memory."""
    def run(env):
        for k in ("BDR_NO_STEP_GATHER", "BDR_NO_MLP_LDS", "BDR_NO_MLP_FUSED"): monkeypatch.delenv(k, raising=False)
        for k in env: monkeypatch.setenv(k, "1")
        rng = np.random.default_rng(9)
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=700, seed=3), (4,), np.float32)
        def push(n):
            rb.push(rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
                    rng.standard_normal((n, 4)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
                    (rng.random(n) < .1).astype(np.int8), np.zeros(n, np.int8))
        push(300)
        a = make_mlp_agent(B, batch_size=32, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=2, n_updates_per_opt=2,
                           double_dqn=True)
        losses = []
        for k in range(12):
            losses.append(a.opt_with_record(rb)["loss"] if k % 4 == 0 else (a.opt(rb), None)[1])
            push(50)   # the ring wraps (capacity 700)
        out = (a.get_params("qnet"), a.get_params("qnet_tgt"), rb.sample_indices(40), losses, a.n_opts)
        a.close(); rb.close()
        return out
    f = run(())
    assert np.isfinite(f[0]).all()
    # separate gather launch; phases exchanging their matrices through global memory instead of LDS; both
    for env in (("BDR_NO_STEP_GATHER",), ("BDR_NO_MLP_LDS",), ("BDR_NO_STEP_GATHER", "BDR_NO_MLP_LDS")):
        s = run(env)
        assert (f[0] == s[0]).all() and (f[1] == s[1]).all(), env
        assert (f[2] == s[2]).all(), env
        assert f[3] == s[3] and f[4] == s[4] == 12, env


@pytest.mark.parametrize("units,Bsz,double", [((256, 256), 64, False), ((128, 96, 64), 200, True), ((64, 64), 128, False)])
def test_mlp_layer_by_layer_path_vs_oracle(B, units, Bsz, double, monkeypatch):
    """Q-networks too large for the one-workgroup step (the reference's own CartPole example is Mlp[256,256] at batch 64,
    examples/gym/dqn_cartpole_tch/src/main.rs:31-46) take the layer-by-layer path on the latency-shaped kernels: z-batched
    32x32-tile forwards, grouped weight gradients, fused reduce + Adam + track.  Agent::opt over the ring against the oracle's
    replay + update, and the same run on the 64x64-tile kernels (BDR_NO_SMALL_GEMM) agrees to rounding; replayed from a hipGraph
    (step_graph.hpp) it is bit-identical to eager launches."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    def run(env):
        for k in ("BDR_NO_SMALL_GEMM", "BDR_STEP_GRAPH", "BDR_NO_MLP_HEAD_FUSE"): monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("BDR_MLP_HEAD_FUSE_WIDE", "1")   # the row-block head also where the heuristic would not pick it (256-wide layers)
        for k, v in env.items(): monkeypatch.setenv(k, v)
        rng = np.random.default_rng(5)
        cap = 4000
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4,), np.float32)
        oref = O.Replay(cap, 42, 16, 8)
        n = 1500
        tr = (rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
              rng.standard_normal((n, 4)).astype(np.float32), np.ones(n, np.float32), (rng.random(n) < .1).astype(np.int8), np.zeros(n, np.int8))
        rb.push(*tr); oref.push(*tr)
        p0 = T.init_params(T.mlp_shapes(4, list(units), 2), 11)
        a = make_mlp_agent(B, units=units, batch_size=Bsz, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1, double_dqn=double)
        a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
        ref = O.DqnOracle(O.mlp_cfg(4, list(units), 2), p0, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1, double_dqn=double)
        for step in range(3):
            rec = a.opt_with_record(rb)
            b = oref.batch(Bsz)
            r = ref.update(b["obs"].view(np.float32).reshape(Bsz, 4), b["act"].view(np.int64).ravel(),
                           b["next_obs"].view(np.float32).reshape(Bsz, 4), b["reward"], b["is_terminated"], probe=True)
            assert rel(a.probe("q_pred_all", Bsz * 2), r["q_pred_all"].ravel()) < 3e-4, step
            assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-7
            assert_grads_close(a.get_params("grad"), r["grads"], T.mlp_shapes(4, list(units), 2), tol=5e-4)
        out = (a.get_params("qnet"), a.get_params("qnet_tgt"), a.get_params("grad"), a.get_params("exp_avg_sq"), np.float32(rec["loss"]),
               a.probe("pred", Bsz), a.probe("tgt", Bsz))
        assert np.abs(out[0] - ref.q).max() < 0.3 * 1e-3 and rel(out[1], ref.q_tgt) < 1e-3
        a.close(); rb.close()
        return out
    lat = run({"BDR_STEP_GRAPH": "0"})
    # the row-block head (last layer + TD rows + loss mean + last dX in one launch, k_mlp_head_td) against the four launches it
    # replaces: parameters, target net, gradients, second moments, the recorded loss and the TD probes bit for bit
    unfused = run({"BDR_STEP_GRAPH": "0", "BDR_NO_MLP_HEAD_FUSE": "1"})
    for x, y in zip(lat, unfused):
        assert (np.asarray(x) == np.asarray(y)).all()
    big = run({"BDR_NO_SMALL_GEMM": "1"})
    assert np.abs(lat[0] - big[0]).max() < 0.3 * 1e-3 and rel(lat[1], big[1]) < 1e-3
    # the same launches replayed from a captured graph (every opt / by the default policy): bit-identical
    for env in ({"BDR_STEP_GRAPH": "1"}, {}):
        g = run(env)
        assert (g[0] == lat[0]).all() and (g[1] == lat[1]).all(), env


@pytest.mark.parametrize("units,Bsz,env", [((64, 64), 32, {}), ((64, 64), 32, {"BDR_NO_MLP_LDS": "1"}), ((64, 64), 32, {"BDR_NO_MLP_FUSED": "1"}),
                                           ((128, 96), 100, {"BDR_STEP_GRAPH": "0"}), ((128, 96), 100, {"BDR_STEP_GRAPH": "0", "BDR_NO_MLP_HEAD_FUSE": "1"})])
def test_mlp_activation_out_on_the_q_network_vs_oracle(B, units, Bsz, env, monkeypatch):
    """MlpConfig::activation_out = true on a DQN Q-network (mlp/base.rs:36: `seq.add_fn(|x| x.relu())` behind the last layer; the
    reference accepts it, rounds 1-3 of this library refused it): Q = relu(z), so rows whose selected Q is clamped to 0 pass no
    gradient.  Every step kernel of the Mlp agent (one-workgroup LDS step, one-workgroup global step, layer-by-layer with the
    row-block head and with the four separate launches) against the C oracle, three updates."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    for k in ("BDR_NO_MLP_LDS", "BDR_NO_MLP_FUSED", "BDR_STEP_GRAPH", "BDR_NO_MLP_HEAD_FUSE", "BDR_NO_SMALL_GEMM"): monkeypatch.delenv(k, raising=False)
    for k, v in env.items(): monkeypatch.setenv(k, v)
    A = 3
    shapes = T.mlp_shapes(4, list(units), A)
    p0 = T.init_params(shapes, 23)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=units, out_dim=A, activation_out=True),
                                                    opt_config=B.OptimizerConfig.Adam(1e-3)),
                      device=0, batch_size=Bsz, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, double_dqn=True)
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    ref = O.DqnOracle(O.mlp_cfg(4, list(units), A, activation_out=True), p0, lr=1e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, double_dqn=True)
    clamped = 0
    for step in range(3):
        rng = np.random.default_rng(900 + step)
        obs = rng.standard_normal((Bsz, 4)).astype(np.float32); nobs = rng.standard_normal((Bsz, 4)).astype(np.float32)
        act = rng.integers(0, A, Bsz); rew = rng.standard_normal(Bsz).astype(np.float32); term = (rng.random(Bsz) < 0.1).astype(np.int8)
        rec = a.update_on_batch(obs, act, nobs, rew, term)
        r = ref.update(obs, act, nobs, rew, term, probe=True)
        q = a.probe("q_pred_all", Bsz * A)
        assert q.min() >= 0.0 and rel(q, r["q_pred_all"].ravel()) < 3e-4, step
        clamped += int((r["pred"] == 0).sum())
        assert rel(a.probe("pred", Bsz), r["pred"]) < 3e-4 and rel(a.probe("tgt", Bsz), r["tgt"]) < 3e-4
        assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-7
        assert_grads_close(a.get_params("grad"), r["grads"], shapes, tol=5e-4)
    assert clamped > 0                     # the masked branch was exercised
    assert np.abs(a.get_params("qnet") - ref.q).max() < 0.3 * 1e-3 and rel(a.get_params("qnet_tgt"), ref.q_tgt) < 1e-3
    assert rel(a.qvalues(obs[:5]), O.net_forward(O.mlp_cfg(4, list(units), A, activation_out=True), a.get_params("qnet"), obs[:5])) < 3e-4
    a.close()


def test_mlp_adamw_matches_aten(B):
    """OptimizerConfig::AdamW (opt.rs:20-27,38-55): decoupled weight decay, custom betas / eps, 5 steps vs ATen."""
    from oracle import torch_ref as T
    shapes = T.mlp_shapes(4, [64, 64], 2)
    kw = dict(beta1=0.8, beta2=0.95, wd=0.05, eps=1e-6)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2),
                                                    opt_config=B.OptimizerConfig.AdamW(2e-3, **kw)),
                      device=0, batch_size=32, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1)
    a = B.Dqn.build(cfg)
    p0 = T.init_params(shapes, 23)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn("mlp", shapes, p0, lr=2e-3, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, adamw=kw)
    for s in range(5):
        batch = _cart(s)
        r = t.update(*batch)
        rec = a.update_on_batch(*batch)
        assert abs(rec["loss"] - r["loss"]) <= QTOL * abs(r["loss"]) + 1e-9
        d = np.abs(a.get_params("qnet").astype(np.float64) - t.params())
        assert d.max() < 0.05 * 2e-3, (s, d.max())          # a small fraction of one optimizer step
    a.close()


@pytest.mark.parametrize("kind", ["mlp", "cnn"])
def test_adamw_amsgrad_matches_aten(B, kind):
    """OptimizerConfig::AdamW{amsgrad: true} (opt.rs:20-27, 45-53): the denominator uses the running maximum of exp_avg_sq.  Batches
    are scaled so the second moment really shrinks between steps (otherwise amsgrad == plain AdamW and the test proves nothing):
    steps vs the ATen restatement (itself checked against torch.optim.AdamW(amsgrad=True) in tests/test_oracle_dqn.py), the
    max_exp_avg_sq arena, and the same run over the replay ring (Agent::opt) stays finite and differs from plain AdamW."""
    from oracle import torch_ref as T
    kw = dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6, amsgrad=True)
    if kind == "mlp":
        shapes, lr = T.mlp_shapes(4, [64, 64], 2), 2e-3
        qc = B.MlpConfig(in_dim=4, units=(64, 64), out_dim=2)
        p0 = T.init_params(shapes, 23)
        batches = [_cart(s) for s in range(6)]
    else:
        shapes, lr = T.cnn_shapes(6), 1e-4
        qc = B.AtariCnnConfig(n_stack=4, out_dim=6)
        p0 = T.init_params(shapes, 5)
        batches = [T.synthetic_atari_batch(8, 6, 70 + s) for s in range(6)]
    # rewards x10 on the first two steps: large gradients first, small ones after -> exp_avg_sq decays below its maximum
    batches = [(o, a_, n, (r * (10.0 if s < 2 else 0.1)).astype(np.float32), t) for s, (o, a_, n, r, t) in enumerate(batches)]
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=qc, opt_config=B.OptimizerConfig.AdamW(lr, **kw)),
                      device=0, batch_size=len(batches[0][3]), critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, arithmetic="f32_exact")
    a = B.Dqn.build(cfg)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn(kind, shapes, p0, lr=lr, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, adamw=kw)
    plain = T.TorchDqn(kind, shapes, p0, lr=lr, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, adamw=dict(kw, amsgrad=False))
    # (the Nature-CNN case runs with arithmetic = f32_exact: at B = 8 ONE ReLU unit at its boundary - which the 1e-6 of the split-operand
    #  forward moves across it now and then, here with the nearest split and not with round 5's truncation split - perturbs every conv
    #  gradient by ~1 % (tools/diag/amsgrad_cnn.py: 2e-5 .. 1e-4 absolute with the split forward, 1e-9 with exact products), and Adam turns a
    #  sign change of a near-zero gradient entry into 2 lr; this test is about the optimizer, which does not depend on the forward's arithmetic)
    for s, batch in enumerate(batches):
        r = t.update(*batch)
        plain.update(*batch)
        rec = a.update_on_batch(*batch)
        assert abs(rec["loss"] - r["loss"]) <= 3e-4 * abs(r["loss"]) + 1e-9, (s, rec["loss"], r["loss"])
        d = np.abs(a.get_params("qnet").astype(np.float64) - t.params())
        assert d.max() < 0.1 * lr, (s, d.max())
    assert a.n_opts == len(batches)
    vmax, v = a.get_params("max_exp_avg_sq"), a.get_params("exp_avg_sq")
    assert (vmax >= v).all() and (vmax > v * 1.5).mean() > 0.1            # the maximum really is ahead of the decayed second moment
    ref_vmax = np.concatenate([x.numpy().ravel() for x in t.vmax])
    assert np.abs(vmax - ref_vmax).max() <= 2e-3 * np.abs(ref_vmax).max()
    # ... and amsgrad is not a no-op here: the restatement with and without it has moved apart by more than the tolerance above
    assert np.abs(t.params() - plain.params()).max() > 0.5 * lr
    a.close()


def test_n_updates_per_opt_and_soft_update_counter(B):
    """opt_ (dqn/base.rs:182-200): n_updates_per_opt critic updates (one batch() each) per opt, ONE soft-update
    counter tick per opt, n_opts += 1 per opt."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    rng = np.random.default_rng(8)
    cap, Bsz = 1000, 16
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=9), (4,), np.float32)
    oref = O.Replay(cap, 9, 16, 8)
    n = 300
    tr = (rng.standard_normal((n, 4)).astype(np.float32), rng.integers(0, 2, (n, 1)).astype(np.int64),
          rng.standard_normal((n, 4)).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32),
          (rng.random(n) < .1).astype(np.int8), np.zeros(n, np.int8))
    rb.push(*tr); oref.push(*tr)
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 31)
    a = make_mlp_agent(B, batch_size=Bsz, lr=1e-3, critic_loss="Mse", tau=0.5, soft_update_interval=2, n_updates_per_opt=3)
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    t = T.TorchDqn("mlp", shapes, p0, lr=1e-3, critic_loss="Mse", tau=0.5, soft_update_interval=10**9)   # soft update driven below
    counter = 0
    for opt in range(4):
        a.opt(rb)
        for _ in range(3):
            b = oref.batch(Bsz)
            t.update(b["obs"].view(np.float32).reshape(Bsz, 4), b["act"].view(np.int64).ravel(),
                     b["next_obs"].view(np.float32).reshape(Bsz, 4), b["reward"], b["is_terminated"])
        counter += 1
        if counter == 2:      # dqn/base.rs:190-194
            counter = 0
            tg = 0.5 * t.params() + 0.5 * t.tgt_params()
            import torch
            with torch.no_grad():
                for d, s_ in zip(t.q_tgt, T.unflatten(tg.astype(np.float32), shapes)):
                    d.copy_(s_)
        assert a.n_opts == opt + 1
    assert rel(a.get_params("qnet"), t.params()) < 2e-4
    assert rel(a.get_params("qnet_tgt"), t.tgt_params()) < 2e-4
    assert rb.sample_indices(4).tolist() == oref.batch(4)["ixs"].tolist()     # 12 batches were drawn on both sides
    a.close(); rb.close()


@pytest.mark.parametrize("Bsz,ddqn,A", [(1, False, 6), (3, True, 6), (33, False, 6), (100, True, 6), (300, False, 6),   # 300: conv1-dW workgroups take two images
                                        (7, True, 4), (3, False, 9), (33, True, 18), (100, False, 33), (65, True, 64)])   # the head kernels' 24- and 64-action blocks
def test_ragged_batch_sizes_vs_oracle(B, Bsz, ddqn, A):
    """Batch sizes that are not multiples of any tile (rows 81*B / 49*B / B, one image per conv1-dW workgroup, two
    rows per head workgroup) and action counts in every register block of the head kernels: Q-values, targets, loss and
    gradients against the C oracle."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(A)
    p0 = T.init_params(shapes, 40 + Bsz)
    obs, act, nobs, rew, term = T.synthetic_atari_batch(Bsz, A, 500 + Bsz)
    term[0] = 1
    a = make_agent(B, A=A, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, double_dqn=ddqn)
    a.set_params(p0, "qnet")
    a.set_params(T.init_params(shapes, 41 + Bsz), "qnet_tgt")
    ref = O.DqnOracle(O.cnn_cfg(A), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, double_dqn=ddqn)
    ref.q_tgt[:] = T.init_params(shapes, 41 + Bsz)
    rec = a.update_on_batch(obs, act, nobs, rew, term)
    r = ref.update(obs, act, nobs, rew, term, probe=True)
    assert rel(a.probe("q_pred_all", Bsz * A), r["q_pred_all"].ravel()) < QTOL
    assert rel(a.probe("pred", Bsz), r["pred"]) < QTOL and rel(a.probe("tgt", Bsz), r["tgt"]) < QTOL
    assert abs(rec["loss"] - r["loss"]) <= QTOL * abs(r["loss"]) + 1e-9
    assert_grads_close(a.get_params("grad"), r["grads"], shapes)
    assert rel(a.qvalues(obs[:1]), O.net_forward(O.cnn_cfg(A), a.get_params("qnet"), obs[:1])) < QTOL
    a.close()


@pytest.mark.parametrize("ns,Bsz,A", [(1, 33, 6), (2, 40, 4), (3, 7, 9), (5, 36, 6), (8, 19, 18)])
def test_other_frame_stack_depths_vs_oracle(B, ns, Bsz, A, tmp_path):
    """AtariCnnConfig::n_stack is a parameter of the reference (cnn/config.rs:14-24, cnn/base.rs:27: conv2d(n_stack, 32, 8, stride 4));
    rounds 1-3 only accepted 4.  conv1's K is 64 * n_stack: the bf16 forward kernel walks 4 * n_stack k-steps, the weight-gradient
    kernel deals 2 * n_stack (channel, kh half) row tiles to its eight waves (idle waves below 4, two tiles per wave above).  One
    update on a fixed minibatch, an opt over a replay buffer of n_stack-frame rows, and the checkpoint's c1.weight shape."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(A, ns)
    p0 = T.init_params(shapes, 60 + ns)
    rng = np.random.default_rng(70 + ns)
    obs = rng.integers(0, 256, (Bsz, ns, 1, 84, 84), dtype=np.uint8); nobs = rng.integers(0, 256, (Bsz, ns, 1, 84, 84), dtype=np.uint8)
    act = rng.integers(0, A, Bsz).astype(np.int64); rew = rng.standard_normal(Bsz).astype(np.float32); term = (rng.random(Bsz) < 0.1).astype(np.int8)
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=ns, out_dim=A), opt_config=B.OptimizerConfig.Adam(1e-4)),
                      device=0, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, double_dqn=True)
    a = B.Dqn.build(cfg)
    assert a.param_count() == sum(int(np.prod(sh)) for sh in shapes)
    a.set_params(p0, "qnet"); a.set_params(T.init_params(shapes, 61 + ns), "qnet_tgt")
    assert (a.get_params("qnet") == p0).all()
    ref = O.DqnOracle(O.cnn_cfg(A, ns), p0, lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, double_dqn=True)
    ref.q_tgt[:] = T.init_params(shapes, 61 + ns)
    rec = a.update_on_batch(obs, act, nobs, rew, term)
    r = ref.update(obs, act, nobs, rew, term, probe=True)
    assert rel(a.probe("q_pred_all", Bsz * A), r["q_pred_all"].ravel()) < QTOL
    assert rel(a.probe("pred", Bsz), r["pred"]) < QTOL and rel(a.probe("tgt", Bsz), r["tgt"]) < QTOL
    assert abs(rec["loss"] - r["loss"]) <= QTOL * abs(r["loss"]) + 1e-9
    assert_grads_close(a.get_params("grad"), r["grads"], shapes)
    assert rel(a.qvalues(obs[:3]), O.net_forward(O.cnn_cfg(A, ns), a.get_params("qnet"), obs[:3])) < QTOL
    # Agent::opt over a ring of n_stack-frame rows (the schedules with two queues), then the checkpoint
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=200, seed=42), (ns, 1, 84, 84), np.uint8)
    rb.fill_synthetic(200, seed=3, kind=0, n_actions=A)
    for _ in range(3):
        a.opt(rb)
    a.sync()
    assert np.isfinite(a.get_params("qnet")).all() and a.n_opts == 4
    a.save_params(str(tmp_path / "m"))
    from border_amd import checkpoint as ck
    names = [("c1.weight", (32, ns, 8, 8)), ("c1.bias", (32,)), ("c2.weight", (64, 32, 4, 4)), ("c2.bias", (64,)), ("c3.weight", (64, 64, 3, 3)),
             ("c3.bias", (64,)), ("l1.weight", (512, 3136)), ("l1.bias", (512,)), ("l2.weight", (A, 512)), ("l2.bias", (A,))]
    t = ck.read(str(tmp_path / "m" / "qnet.pt.tch"), names)                 # the reference's variable names, c1.weight [32][n_stack][8][8]
    assert t["c1.weight"].shape == (32, ns, 8, 8)
    assert (np.concatenate([t[n].ravel() for n, _ in names]) == a.get_params("qnet")).all()
    b = B.Dqn.build(cfg)
    b.load_params(str(tmp_path / "m"))
    assert (b.get_params("qnet") == a.get_params("qnet")).all()
    with pytest.raises(B.BdrError, match="do not match the AtariCnn input"):   # rows of another depth are refused, not misread
        wrong = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=20, seed=1), (ns + 1, 1, 84, 84), np.uint8)
        wrong.fill_synthetic(20, seed=1, kind=0, n_actions=A)
        a.opt(wrong)
    a.close(); b.close(); rb.close()


def test_checkpoints_are_safetensors_with_reference_names(B, tmp_path):
    """save_params / load_params (dqn/base.rs:345-371) use the safetensors container tch's VarStore reads and writes
    for *.safetensors paths: the official `safetensors` package must read what the library wrote (reference variable
    names, OIHW / [out,in] layouts) and the library must load what the package wrote (any key order, extra metadata)."""
    from safetensors.numpy import load_file, save_file
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(6)
    names = ["c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias", "l1.weight", "l1.bias", "l2.weight", "l2.bias"]
    a = make_agent(B, batch_size=4)
    p0, p1 = T.init_params(shapes, 61), T.init_params(shapes, 62)
    a.set_params(p0, "qnet"); a.set_params(p1, "qnet_tgt")
    a.set_checkpoint_format("safetensors")
    files = a.save_params(str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["qnet.safetensors", "qnet_tgt.safetensors"]
    for f, p in zip(files, (p0, p1)):
        d = load_file(f)
        assert sorted(d) == sorted(names)
        o = 0
        for nm, sh in zip(names, shapes):
            n = int(np.prod(sh))
            assert d[nm].dtype == np.float32 and d[nm].shape == tuple(sh)
            assert (d[nm].ravel() == p[o:o + n]).all(), nm
            o += n
    # the other direction: files produced by the official writer (sorted keys, metadata entry)
    q0, q1 = T.init_params(shapes, 63), T.init_params(shapes, 64)
    for f, q in zip(files, (q0, q1)):
        o, d = 0, {}
        for nm, sh in zip(names, shapes):
            n = int(np.prod(sh))
            d[nm] = q[o:o + n].reshape(sh).copy()
            o += n
        save_file(d, f, metadata={"format": "pt"})
    a.load_params(str(tmp_path))
    assert (a.get_params("qnet") == q0).all() and (a.get_params("qnet_tgt") == q1).all()
    # a file with a wrong shape is rejected
    bad = {nm: np.zeros(sh, np.float32) for nm, sh in zip(names, shapes)}
    bad["l2.bias"] = np.zeros(7, np.float32)
    save_file(bad, files[0])
    with pytest.raises(B.BdrError):
        a.load_params(str(tmp_path))
    a.close()


def test_checkpoints_default_to_the_reference_pt_tch_archives(B, tmp_path):
    """By default save_params writes the reference's files - `qnet.pt.tch`, `qnet_tgt.pt.tch` (dqn/base.rs:348-356) - in
    the container tch's VarStore::save produces for them (libtorch named-tensor archive).  libtorch's TorchScript loader,
    which is what VarStore::load calls, must see the reference's variable names / layouts; and a Nature-CNN exported from
    PyTorch (torch.jit.save of a module with submodules c1..l2) must load into the agent and compute the same Q-values."""
    import torch
    from oracle import torch_ref as T
    shapes = T.cnn_shapes(6)
    names = ["c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias", "l1.weight", "l1.bias", "l2.weight", "l2.bias"]
    a = make_agent(B, batch_size=4)
    p0, p1 = T.init_params(shapes, 71), T.init_params(shapes, 72)
    a.set_params(p0, "qnet"); a.set_params(p1, "qnet_tgt")
    files = a.save_params(str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["qnet.pt.tch", "qnet_tgt.pt.tch"]
    for f, p in zip(files, (p0, p1)):
        got = list(torch.jit.load(f).named_parameters())
        assert [n for n, _ in got] == names
        o = 0
        for (n, t), sh in zip(got, shapes):
            k = int(np.prod(sh))
            assert t.dtype == torch.float32 and tuple(t.shape) == tuple(sh)
            assert (t.detach().numpy().ravel() == p[o:o + k]).all(), n
            o += k
    b = make_agent(B, batch_size=4, param_seed=9)
    b.load_params(str(tmp_path))
    assert (b.get_params("qnet") == p0).all() and (b.get_params("qnet_tgt") == p1).all()
    b.close()

    # Python-side export of the same architecture (cnn/base.rs:23-36)
    class AtariCnn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = torch.nn.Conv2d(4, 32, 8, 4)
            self.c2 = torch.nn.Conv2d(32, 64, 4, 2)
            self.c3 = torch.nn.Conv2d(64, 64, 3, 1)
            self.l1 = torch.nn.Linear(3136, 512)
            self.l2 = torch.nn.Linear(512, 6)

        def forward(self, x):
            x = x.squeeze(2).float() / 255.0
            x = torch.relu(self.c3(torch.relu(self.c2(torch.relu(self.c1(x))))))
            return self.l2(torch.relu(self.l1(x.flatten(1))))

    torch.manual_seed(11)
    net = AtariCnn()
    exported = tmp_path / "exported"
    exported.mkdir()
    scripted = torch.jit.script(net)
    torch.jit.save(scripted, str(exported / "qnet.pt.tch"))
    torch.jit.save(scripted, str(exported / "qnet_tgt.pt.tch"))
    a.load_params(str(exported))
    obs = np.random.default_rng(5).integers(0, 256, (7, 4, 1, 84, 84), dtype=np.uint8)
    with torch.no_grad():
        want = net(torch.from_numpy(obs)).numpy()
    assert rel(a.qvalues(obs), want) < QTOL
    # only the other container present: load falls back to it
    only_st = tmp_path / "only_st"
    a.set_checkpoint_format("safetensors")
    a.save_params(str(only_st))
    a.set_checkpoint_format("tch")
    c = make_agent(B, batch_size=4, param_seed=10)
    c.load_params(str(only_st))
    assert (c.get_params("qnet") == a.get_params("qnet")).all()
    c.close(); a.close()


def test_opt_stream_is_deterministic_and_overlap_invariant(B):
    """200 opt steps over the same synthetic ring from the same initial parameters: bit-identical parameters between
    two runs, and between every backward schedule (BDR_SCHED: 0 serial, 1 two streams ordered by events, 2 any-order
    launches, 3 = default, two streams ordered by device flags and gate kernels) - every reduction has a fixed order, so
    any difference would be a race between the queues."""
    def run(sched):
        if sched is None:
            os.environ.pop("BDR_SCHED", None)
        else:
            os.environ["BDR_SCHED"] = str(sched)
        try:
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=20_000, seed=42), (4, 1, 84, 84), "uint8")
            rb.fill_synthetic(20_000, seed=3, kind=0, n_actions=6)
            a = make_agent(B, batch_size=64, critic_loss="SmoothL1", tau=1.0, soft_update_interval=50, param_seed=5)
            a.train()
            for _ in range(200):
                a.opt(rb)
            a.sync()
            p, t = a.get_params("qnet"), a.get_params("qnet_tgt")
            a.close(); rb.close()
            return p, t
        finally:
            os.environ.pop("BDR_SCHED", None)

    p1, t1 = run(None)
    assert np.isfinite(p1).all()
    for sched in (None, 0, 1, 2, 3):
        p, t = run(sched)
        assert (p1 == p).all() and (t1 == t).all(), sched
    # the default puts the gather + target forward on the agent's third queue; BDR_TQ=0 keeps them on the weight-gradient queue
    os.environ["BDR_TQ"] = "0"
    try:
        p, t = run(None)
    finally:
        os.environ.pop("BDR_TQ", None)
    assert (p1 == p).all() and (t1 == t).all(), "BDR_TQ=0"


def test_mixed_api_sequences_on_the_flag_ordered_schedule(B, tmp_path):
    """Entry points that touch the parameters or the batch buffers between opt() calls (schedule 3 keeps work in flight on a
    second queue and gathers the next batch into an alternate buffer set): every sequence must equal the serial schedule's
    result bit for bit, a buffer must outlive the agent that last gathered from it, and two agents may share one buffer."""
    rng = np.random.default_rng(3)

    def run(sched):
        os.environ["BDR_SCHED"] = str(sched)
        try:
            rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=300, seed=7), (4, 1, 84, 84), "uint8")
            rb.fill_synthetic(200, seed=1, kind=0, n_actions=6)
            a = make_agent(B, batch_size=16, critic_loss="SmoothL1", tau=0.5, soft_update_interval=3, param_seed=6)
            b = make_agent(B, batch_size=8, critic_loss="Mse", tau=1.0, soft_update_interval=2, param_seed=8)
            a.train(); b.train()
            r2 = np.random.default_rng(11)
            out = []
            for it in range(12):
                a.opt(rb)
                if it % 3 == 0:
                    b.opt(rb)                                   # a second consumer of the same buffer
                if it % 4 == 1:                                # host pushes between opts (wrap after 100 more rows)
                    n = 30
                    rb.push(r2.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8), r2.integers(0, 6, (n, 1)).astype(np.int64),
                            r2.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8), r2.standard_normal(n).astype(np.float32),
                            (r2.random(n) < 0.1).astype(np.int8), np.zeros(n, np.int8))
                if it % 5 == 2:
                    out.append(a.qvalues(r2.integers(0, 256, (3, 4, 1, 84, 84), dtype=np.uint8)).copy())
                if it == 6:
                    p = a.get_params("qnet")
                    a.set_params(p * np.float32(0.999), "qnet")
                if it == 8:
                    a.save_params(str(tmp_path / f"s{sched}"))
                    a.load_params(str(tmp_path / f"s{sched}"))
                if it == 9:
                    rec = a.opt_with_record(rb)
                    out.append(np.float32(rec["loss"]))
                if it == 10:                                    # explicit minibatch of another size
                    m = 5
                    a.update_on_batch(r2.integers(0, 256, (m, 4, 1, 84, 84), dtype=np.uint8), r2.integers(0, 6, m).astype(np.int64),
                                      r2.integers(0, 256, (m, 4, 1, 84, 84), dtype=np.uint8), r2.standard_normal(m).astype(np.float32),
                                      np.zeros(m, np.int8))
            a.sync(); b.sync()
            out += [a.get_params("qnet"), a.get_params("qnet_tgt"), b.get_params("qnet")]
            a.close(); b.close()
            # the buffer outlives the agents whose (destroyed) queue gathered from it last
            rb.push(r2.integers(0, 256, (2, 4, 1, 84, 84), dtype=np.uint8), np.zeros((2, 1), np.int64),
                    r2.integers(0, 256, (2, 4, 1, 84, 84), dtype=np.uint8), np.zeros(2, np.float32), np.zeros(2, np.int8), np.zeros(2, np.int8))
            g = rb.batch(4)
            out += [g.ix_sample.copy(), g.obs.copy()]
            rb.close()
            return out
        finally:
            os.environ.pop("BDR_SCHED", None)

    ref = run(0)
    for sched in (3, 1):
        got = run(sched)
        assert len(got) == len(ref)
        for k, (x, y) in enumerate(zip(got, ref)):
            assert np.array_equal(np.asarray(x), np.asarray(y)), (sched, k)


def test_shared_hardware_queue_is_detected_and_falls_back(B, tmp_path):
    """HIP multiplexes streams onto a bounded pool of hardware queues.  The flag-ordered schedule needs the agent's two streams
    on different queues (a gate kernel sharing an in-order queue with its producer would wait for a kernel behind it); the
    agent checks that at creation.  With the pool forced to 2 queues (buffer stream + one more) the check must fail, the
    agent must say so, run the serial schedule instead of hanging, and produce the same parameters bit for bit; with 3 the
    prioritized-replay queue aliases and its dependency falls back to an event."""
    import subprocess
    import sys
    script = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import border_amd as B
per = sys.argv[1] == "per"
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=2000, seed=42, per_config=B.PerConfig(n_opts_final=30) if per else None),
                          (4, 1, 84, 84), "uint8")
rb.fill_synthetic(2000, seed=3, kind=0, n_actions=6)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  device=0, batch_size=32, critic_loss="SmoothL1", tau=1.0, soft_update_interval=20, param_seed=5)
a = B.Dqn.build(cfg); a.train()
for _ in range(40): a.opt(rb)
a.sync()
np.save(sys.argv[2], a.get_params("qnet"))
""" % os.path.join(os.path.dirname(__file__), "..")
    outs = {}
    for per in ("uniform", "per"):
        for q in ("default", "2", "3"):
            env = dict(os.environ)
            env.pop("BDR_SCHED", None)
            if q != "default":
                env["GPU_MAX_HW_QUEUES"] = q
            f = str(tmp_path / f"{per}_{q}.npy")
            r = subprocess.run([sys.executable, "-c", script, per, f], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-800:]
            outs[(per, q)] = (np.load(f), r.stderr)
        ref, _ = outs[(per, "default")]
        for q in ("2", "3"):
            got, err = outs[(per, q)]
            assert np.array_equal(got, ref), (per, q)
            assert ("share a hardware queue" in err) == (q == "2"), (per, q, err[-300:])


CNN_VARS = [("c1.weight", (32, 4, 8, 8)), ("c1.bias", (32,)), ("c2.weight", (64, 32, 4, 4)), ("c2.bias", (64,)),
            ("c3.weight", (64, 64, 3, 3)), ("c3.bias", (64,)), ("l1.weight", (512, 3136)), ("l1.bias", (512,)),
            ("l2.weight", (6, 512)), ("l2.bias", (6,))]


@pytest.mark.parametrize("net", ["cnn", "mlp"])
def test_opt_with_record_verbose_keys_param_stats_and_ratio_best_act(B, net):
    """Agent::opt_with_record with record_verbose_level >= 2 (dqn/base.rs:316-342): the update's own scalars, then
    qnet.param_stats() - `<var>_mean` / `<var>_std` (population std, util.rs:64-80) of every variable, computed from the
    parameters AFTER the step - and ratio_best_act = n_samples_best_act / n_samples_act, which resets both counters; a plain
    update (update_on_batch) carries only the five update scalars; level 0 only "loss"."""
    import torch
    if net == "cnn":
        variables, obs_shape, obs_dtype, A = CNN_VARS, (4, 1, 84, 84), np.uint8, 6
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=500, seed=1), obs_shape, obs_dtype)
        rb.fill_synthetic(500, seed=2, kind=0, n_actions=A)
        a = make_agent(B, A=A, batch_size=16, record_verbose_level=2, critic_loss="SmoothL1")
        obs = np.random.default_rng(0).integers(0, 256, (3,) + obs_shape, dtype=np.uint8)
    else:
        variables, A = [(f"mlp.ln{i}.{k}", s) for i, (o, n) in enumerate([(64, 4), (64, 64), (2, 64)]) for k, s in (("weight", (o, n)), ("bias", (o,)))], 2
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=500, seed=1), (4,), np.float32)
        rb.fill_synthetic(500, seed=2, kind=1, n_actions=A)
        a = make_mlp_agent(B, batch_size=16, record_verbose_level=2)
        obs = np.random.default_rng(0).standard_normal((3, 4)).astype(np.float32)
    a.train()
    a.set_explorer(B.EpsilonGreedy(final_step=10), seed=3)
    infos = [a.sample(obs, return_info=True)[1] for _ in range(12)]
    assert infos[-1]["n_samples_act"] == 12
    want_ratio = infos[-1]["n_samples_best_act"] / 12.0
    rec = a.opt_with_record(rb)
    keys = list(rec)
    assert keys[:5] == ["loss", "pred_mean", "reward_mean", "tgt_mean", "tgt_minus_pred_mean"]
    assert keys[5:-1] == [f"{v}_{s}" for v, _ in variables for s in ("mean", "std")] and keys[-1] == "ratio_best_act"
    p = a.get_params("qnet")
    o = 0
    for v, shp in variables:
        n = int(np.prod(shp))
        t = torch.from_numpy(p[o:o + n].copy())
        assert abs(rec[f"{v}_mean"] - float(t.mean())) <= 1e-6 * max(1.0, abs(float(t.mean()))) + 1e-9, v
        assert abs(rec[f"{v}_std"] - float(t.std(unbiased=False))) <= 1e-5 * float(t.std(unbiased=False)) + 1e-9, v
        o += n
    assert o == p.size
    assert abs(rec["ratio_best_act"] - np.float32(want_ratio)) < 1e-7
    assert abs(rec["tgt_minus_pred_mean"] - (rec["tgt_mean"] - rec["pred_mean"])) < 1e-5
    # the counters were reset (dqn/base.rs:337-338): no samples since -> ratio 0
    assert a.sample(obs, return_info=True)[1]["n_samples_act"] == 1
    a.sample(obs)
    rec2 = a.opt_with_record(rb)
    assert 0.0 <= rec2["ratio_best_act"] <= 1.0 and a.sample(obs, return_info=True)[1]["n_samples_act"] == 1
    assert a.opt_with_record(rb)["ratio_best_act"] in (0.0, 1.0)
    # the C struct entry point still reports the five update scalars
    import ctypes as C
    from border_amd import _lib
    r = _lib.DqnRecordC()
    _lib.check(_lib.lib().bdr_agent_opt_with_record(a.handle, rb.handle, C.byref(r)))
    assert r.has_verbose == 1 and np.isfinite(r.loss)
    a.close(); rb.close()


def test_out_of_range_action_is_flagged_not_read_out_of_bounds(B):
    """update_critic gathers Q(s, a) with the stored action (dqn/base.rs:71-74); the reference's gather raises for an index
    outside [0, A).  Here the kernel clamps the index (no out-of-bounds read) and raises a device flag that the next
    synchronising call reports; the agent stays usable."""
    from oracle import torch_ref as T
    a = make_agent(B, A=6, batch_size=8)
    obs, act, nobs, rew, term = T.synthetic_atari_batch(8, 6, 5)
    act = np.array(act, np.int64); act[3] = 6
    with pytest.raises(B.BdrError) as e:
        a.update_on_batch(obs, act, nobs, rew, term)
    assert e.value.code == 1 and "action" in str(e.value)
    act[3] = -1
    with pytest.raises(B.BdrError):
        a.update_on_batch(obs, act, nobs, rew, term)
    act[3] = 5
    assert np.isfinite(a.update_on_batch(obs, act, nobs, rew, term)["loss"])
    a.sync()
    m = make_mlp_agent(B, batch_size=8)
    rng = np.random.default_rng(0)
    mo, mn = rng.standard_normal((8, 4)).astype(np.float32), rng.standard_normal((8, 4)).astype(np.float32)
    with pytest.raises(B.BdrError) as e:
        m.update_on_batch(mo, np.array([0, 1, 2, 0, 1, 0, 1, 0]), mn, rew, term)
    assert e.value.code == 1
    assert np.isfinite(m.update_on_batch(mo, np.array([0, 1, 1, 0, 1, 0, 1, 0]), mn, rew, term)["loss"])
    a.close(); m.close()


def test_save_params_creates_the_directory_like_the_reference(B, tmp_path):
    """Agent::save_params calls fs::create_dir_all(path) first (dqn/base.rs:346): a C / Rust caller of bdr_agent_save_params
    may pass a directory that does not exist yet."""
    from border_amd import _lib
    a = make_mlp_agent(B, batch_size=4)
    d = tmp_path / "new" / "nested" / "dir"
    _lib.check(_lib.lib().bdr_agent_save_params(a.handle, str(d).encode()))
    assert sorted(os.listdir(d)) == ["qnet.pt.tch", "qnet_tgt.pt.tch"]
    f = tmp_path / "a_file"
    f.write_text("x")
    with pytest.raises(B.BdrError) as e:
        _lib.check(_lib.lib().bdr_agent_save_params(a.handle, str(f / "sub").encode()))
    assert e.value.code == 5
    a.close()


def test_gate_timeout_poisons_the_step_reports_and_recovers(B, tmp_path):
    """Schedule 3 orders the two backward queues with spin-wait gate kernels.  Forced failure: the queue check is skipped
    (BDR_SKIP_QUEUE_CHECK) while the hardware-queue pool is 2, so a gate sits in front of its own producer and must time out
    (20 ms limit).  Required behaviour: no hang; parameter-writing kernels behind the failed gate are skipped (the
    parameters after the failed steps are the initial ones, bit for bit); the failure is reported - by bdr_agent_sync and by
    the asynchronous poll of a loop that only calls Agent::opt - and cleared; the agent then continues with event ordering
    and trains normally."""
    import subprocess
    import sys
    script = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import border_amd as B
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=2000, seed=42), (4, 1, 84, 84), "uint8")
rb.fill_synthetic(2000, seed=3, kind=0, n_actions=6)
cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                  device=0, batch_size=32, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, param_seed=5)
a = B.Dqn.build(cfg); a.train()
p0 = a.get_params("qnet")
mode = sys.argv[1]
if mode == "sync":
    for _ in range(3): a.opt(rb)
    try:
        a.sync(); print("NO_ERROR")
    except B.BdrError as e:
        print("ERR", e.code, "gate" in str(e))
else:          # a loop that never synchronises: the poll inside Agent::opt must surface the failure
    seen = None
    for k in range(1500):
        try:
            a.opt(rb)
        except B.BdrError as e:
            seen = (k, e.code, "gate" in str(e)); break
    print("ERR" if seen else "NO_ERROR", *(seen or ()))
print("SAME", bool((a.get_params("qnet") == p0).all()))
a.sync()                                   # clean again
print("NOPTS", a.n_opts)                   # the skipped updates are not counted: n_opts / the Adam step number are those of the state on the device
for _ in range(5): a.opt(rb)
rec = a.opt_with_record(rb)
print("LOSS", np.isfinite(rec["loss"]), "MOVED", bool((a.get_params("qnet") != p0).any()))
""" % os.path.join(os.path.dirname(__file__), "..")
    for mode in ("sync", "poll"):
        env = dict(os.environ)
        env.pop("BDR_SCHED", None)
        env.update(GPU_MAX_HW_QUEUES="2", BDR_SKIP_QUEUE_CHECK="1", BDR_GATE_LIMIT_MS="20")
        r = subprocess.run([sys.executable, "-c", script, mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        out = r.stdout.split("\n")
        assert out[0].startswith("ERR") and " 3 " in out[0] + " " and "True" in out[0], (mode, r.stdout, r.stderr[-500:])
        assert out[1] == "SAME True", (mode, r.stdout)
        assert out[2] in ("NOPTS 0", "NOPTS 1") and (mode == "poll" or out[2] == "NOPTS 0"), (mode, r.stdout)
        assert out[3] == "LOSS True MOVED True", (mode, r.stdout)
        assert "continues with event ordering" in r.stderr and "were rolled back" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
def test_a_deferred_report_is_told_apart_from_a_failure_of_the_call_itself(B):
    """`bdr_last_error_is_deferred` (include/border_amd.h): Agent::opt returns () in the reference, so the Rust shim needs to know whether a
    failing bdr_agent_opt reports an EARLIER step's device-side condition (log, keep, enqueue again) or failed on its own arguments (the
    reference's panic).  An empty buffer is the call's own failure; an action index outside [0, n_actions) that reached a TD step is a
    deferred report, surfaced by the next synchronising call, after which the agent trains on."""
    from border_amd import _lib
    L = _lib.lib()
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=64, seed=1), (4, 1, 84, 84), "uint8")
    cfg = B.DqnConfig(model_config=B.DqnModelConfig(q_config=B.AtariCnnConfig(n_stack=4, out_dim=6), opt_config=B.OptimizerConfig.Adam(1e-4)),
                      device=0, batch_size=8, param_seed=2)
    a = B.Dqn.build(cfg); a.train()
    with pytest.raises(B.BdrError) as e:
        a.opt(rb)                                      # empty buffer: BDR_ERR_EMPTY, the call's own failure
    assert e.value.code == 4 and L.bdr_last_error_is_deferred() == 0
    g = np.random.default_rng(0)
    n = 32
    obs = g.integers(0, 256, (n, 4, 1, 84, 84), dtype=np.uint8)
    act = np.full((n, 1), 17, np.int64)                # no such action
    rb.push(obs, act, obs, np.zeros(n, np.float32), np.zeros(n, np.int8), np.zeros(n, np.int8))
    a.opt(rb)                                          # enqueued: the TD kernel clamps the index and raises the device flag
    with pytest.raises(B.BdrError) as e:
        a.sync()
    assert e.value.code == 1 and "action index" in str(e.value) and L.bdr_last_error_is_deferred() == 1
    a.sync()                                           # reported once, cleared
    # ... and the same condition raised by the step the failing call ITSELF ran (Agent::opt_with_record: enqueue, synchronise, check): kind 2.
    # The step has run exactly once and its record is in the caller's buffer - a caller that "retried" would take a second optimizer step
    # (the Rust shim did).
    import ctypes as C
    n0 = a.n_opts
    vals, n_out = (C.c_float * 64)(*([float("nan")] * 64)), C.c_int32(-1)
    rc = L.bdr_agent_opt_with_scalars(a.handle, rb.handle, vals, 64, C.byref(n_out))
    assert rc == 1 and L.bdr_last_error_is_deferred() == 2 and b"action index" in L.bdr_last_error()
    assert a.n_opts == n0 + 1 and n_out.value >= 1 and np.isfinite(vals[0])          # one step, and its loss is there
    a.sync()
    assert a.n_opts == n0 + 1
    a.close(); rb.close()


@pytest.mark.gpu
def test_split_operand_conv_planes_follow_every_parameter_writer(B, monkeypatch, tmp_path):
    """conv2 / conv3 (forward and input gradients) run on the bf16 matrix cores from bf16 planes of W2 / W3 kept beside each parameter set
    (csrc/dqn.hip: cpl, k_reduce_adam writes the online planes with the parameters; soft updates, set_params, load, the all-reduce
    and a gate time-out leave a set stale and the next forward re-splits it).  A stale plane would be a whole Adam step / soft update
    behind: at lr = 3e-3, tau = 0.5 that is tens of percent on the loss.  A twin agent on the exact FP32-MFMA kernels
    (bdr_dqn_config::arithmetic = BDR_ARITH_F32_EXACT) over the same ring and the same parameters must agree at every step within the split's own error class
    (six of nine partial products: ~2e-6 per layer) plus Adam's amplification of it; and the acting kernels (exact f32, n <= 8)
    must agree with the training forward (planes) on the same rows after the run."""
    from oracle import torch_ref as T
    cap, Bsz = 256, 32
    p0 = T.init_params(T.cnn_shapes(6), 9)
    kw = dict(batch_size=Bsz, lr=3e-3, critic_loss="SmoothL1", tau=0.5, soft_update_interval=3)
    agents, bufs = [], []
    for exact in (False, True):
        monkeypatch.delenv("BDR_DQN_F32_EXACT", raising=False)
        a = make_agent(B, arithmetic="f32_exact" if exact else "bf16x3_6", **kw)   # bdr_dqn_config::arithmetic, not the A/B variable
        a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4, 1, 84, 84), np.uint8)
        rb.fill_synthetic(cap, seed=3, kind=0, n_actions=6)
        agents.append(a); bufs.append(rb)
    monkeypatch.delenv("BDR_DQN_F32_EXACT", raising=False)
    rng = np.random.default_rng(4)
    obs = rng.integers(0, 256, (24, 4, 1, 84, 84), dtype=np.uint8)

    def both(step):
        recs = [a.opt_with_record(rb) for a, rb in zip(agents, bufs)]
        assert abs(recs[0]["loss"] - recs[1]["loss"]) <= 2e-3 * abs(recs[1]["loss"]) + 1e-7, (step, recs[0]["loss"], recs[1]["loss"])

    def acting_agrees(a):
        q_train = a.qvalues(obs)                    # 24 rows: the training forward (planes of the online set)
        q_act = a.qvalues(obs[:8])                  # 8 rows: the acting kernels (f32 weights)
        assert np.abs(q_act - q_train[:8]).max() <= 1e-5 * np.abs(q_train).max()

    for step in range(7): both(step)                # two soft updates inside
    acting_agrees(agents[0])
    # writers other than the update itself: set_params on both sets, a checkpoint load, the raw arena pointer
    pq, pt = agents[1].get_params("qnet"), agents[1].get_params("qnet_tgt")
    for a in agents: a.set_params(pt, "qnet"); a.set_params(pq, "qnet_tgt")
    acting_agrees(agents[0])
    for step in range(7, 10): both(step)
    agents[1].save_params(str(tmp_path / "ck"))
    for a in agents: a.load_params(str(tmp_path / "ck"))
    acting_agrees(agents[0])
    for step in range(10, 13): both(step)
    agents[0].arena_device_ptr("qnet"); agents[0].arena_device_ptr("qnet_tgt")    # from here on: re-split before every forward
    for step in range(13, 17): both(step)
    acting_agrees(agents[0])
    for a, rb in zip(agents, bufs): a.close(); rb.close()


def test_split_forward_error_budget_per_layer_at_c2_shapes(B, monkeypatch):
    """The split-operand forward's OWN error budget, layer by layer, at the headline shapes (B = 256: conv2 = [20736][512] x [512][64],
    conv3 = [12544][576] x [576][64]).  Each layer's output is compared with an f64 evaluation of that layer on the device's own
    input activations (probes 5-7), so only the layer's arithmetic shows: products (exact f32 on the FP32 MFMA, or six of the nine bf16
    partial products of round-to-nearest split operands) and the f32 accumulation order.  Budget, relative to the layer's largest
    output: 1e-6 for either arithmetic (an f32 dot product of K = 512 / 576 terms in any order is ~4e-7; the three dropped
    products add <= 3 * 2^-26 per product before cancellation; measured on MI355X: 8.5e-7 / 7.9e-7 split, 8.0e-7 / 4.9e-7 exact, mean signed
    -5e-9 / -6e-9 split, 7e-10 / -3e-10 exact), the two arithmetics within 2e-6 of each other (measured 1.2e-6), and the split path's
    MEAN signed error against f64 below 2e-8: the nearest split leaves no bias (the truncation split of rounds 4-5 pushed
    every dropped product the same way).  The env override is also covered: BDR_DQN_F32_EXACT=0 forces the split kernels onto an
    agent configured exact, and the labels / results say so."""
    import torch
    from oracle import torch_ref as T
    monkeypatch.delenv("BDR_DQN_F32_EXACT", raising=False)
    Bsz, A = 256, 6
    p0 = T.init_params(T.cnn_shapes(A), 12)
    obs, act, nobs, rew, term = T.synthetic_atari_batch(Bsz, A, 77)
    w = T.unflatten(p0, T.cnn_shapes(A))     # c1.w c1.b c2.w c2.b c3.w c3.b l1.w l1.b l2.w l2.b
    out = {}
    for mode in ("bf16x3_6", "f32_exact"):
        a = make_agent(B, A, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, arithmetic=mode)
        a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
        a.update_on_batch(obs, act, nobs, rew, term)
        a1 = a.probe("act_conv1", Bsz * 400 * 32).reshape(Bsz, 20, 20, 32)
        a2 = a.probe("act_conv2", Bsz * 81 * 64).reshape(Bsz, 9, 9, 64)
        a3 = a.probe("act_conv3", Bsz * 49 * 64).reshape(Bsz, 7, 7, 64)
        out[mode] = (a1, a2, a3)
        a.close()

    def layer64(x_nhwc, wt, bias, stride):
        x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).double()
        y = torch.nn.functional.conv2d(x, wt.detach().double(), bias.detach().double(), stride=stride).relu()
        return y.permute(0, 2, 3, 1).numpy()

    assert (out["bf16x3_6"][0] == out["f32_exact"][0]).all()            # conv1 is the same kernel in both
    errs = {}
    for mode, (a1, a2, a3) in out.items():
        r2, r3 = layer64(a1, w[2], w[3], 2), layer64(a2, w[4], w[5], 1)
        e2, e3 = (a2.astype(np.float64) - r2) / np.abs(r2).max(), (a3.astype(np.float64) - r3) / np.abs(r3).max()
        errs[mode] = (np.abs(e2).max(), np.abs(e3).max(), e2[r2 > 0].mean(), e3[r3 > 0].mean())
        assert np.abs(e2).max() < 1e-6 and np.abs(e3).max() < 1e-6, (mode, errs[mode])
    print("per-layer error vs f64 (max conv2, max conv3, mean signed conv2, mean signed conv3):", errs)
    assert abs(errs["bf16x3_6"][2]) < 2e-8 and abs(errs["bf16x3_6"][3]) < 2e-8, errs
    for i in (1, 2):
        d = np.abs(out["bf16x3_6"][i].astype(np.float64) - out["f32_exact"][i]).max() / np.abs(out["f32_exact"][i]).max()
        assert d < 2e-6, (i, d)
    # the A/B variable overrides the field in both directions
    monkeypatch.setenv("BDR_DQN_F32_EXACT", "0")
    a = make_agent(B, A, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, arithmetic="f32_exact")
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    a.update_on_batch(obs, act, nobs, rew, term)
    assert (a.probe("act_conv2", Bsz * 81 * 64).reshape(Bsz, 9, 9, 64) == out["bf16x3_6"][1]).all()
    a.close()
    monkeypatch.setenv("BDR_DQN_F32_EXACT", "1")
    a = make_agent(B, A, batch_size=Bsz, critic_loss="SmoothL1", tau=1.0, soft_update_interval=10000, arithmetic="bf16x3_6")
    a.set_params(p0, "qnet"); a.set_params(p0, "qnet_tgt")
    a.update_on_batch(obs, act, nobs, rew, term)
    assert (a.probe("act_conv2", Bsz * 81 * 64).reshape(Bsz, 9, 9, 64) == out["f32_exact"][1]).all()
    a.close()
