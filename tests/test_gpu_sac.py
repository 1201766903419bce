"""HIP SAC step (through the C ABI) vs the committed PyTorch-autograd goldens and the C oracle.
BASELINE config 5 shape: obs 17 / act 6, twin-Q, Auto entropy coefficient."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _opt(B, lr, adamw):
    """OptimizerConfig of one model: Adam{lr} or AdamW{lr, ..} (opt.rs:13-28)"""
    return None if adamw is None else B.OptimizerConfig.AdamW(lr, **adamw)


def _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0):
    cfg = B.SacConfig(obs_dim=od, act_dim=ad, pi_units=tuple(pu), q_units=tuple(qu), lr_actor=kw["lr_actor"], lr_critic=kw["lr_critic"],
                      opt_actor=_opt(B, kw["lr_actor"], kw.get("adamw_actor")), opt_critic=_opt(B, kw["lr_critic"], kw.get("adamw_critic")),
                      ent_coef_mode=kw["ent_coef"], critic_loss=kw["critic_loss"], reward_scale=kw.get("reward_scale", 1.0),
                      n_critics=nc, batch_size=Bsz, device=0)
    a = B.Sac.build(cfg)
    a.set_params(pi0, "pi")
    for i in range(nc):
        a.set_params(q0[i], f"qnet_{i}"); a.set_params(q0[i], f"qnet_tgt_{i}")
    return a


def _run(B, name, golden_dir):
    from oracle import torch_ref as T
    od, ad, pu, qu, nc, Bsz, steps, kw, pi0, q0, seed = MG.sac_case_params(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    a = _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0)
    assert (a.get_params("pi") == pi0).all() and (a.get_params("qnet_0") == q0[0]).all()
    for s in range(steps):
        rec = a.update_on_batch(*MG.sac_case_batch(name, s))
        # log_p is ill-conditioned near |a| -> 1 (see tests/test_oracle_sac.py): 5e-4 on the scalars
        for k in ("loss_critic", "loss_actor", "ent_coef"):
            assert abs(rec[k] - g[f"s{s}_{k}"]) <= 5e-4 * abs(g[f"s{s}_{k}"]) + 1e-6, (s, k, rec[k], g[f"s{s}_{k}"])
        assert rel(a.get_params("pi", "grad"), g[f"s{s}_pi_grads"]) < 2e-3, (s, rel(a.get_params("pi", "grad"), g[f"s{s}_pi_grads"]))
        for i in range(nc):
            assert rel(a.get_params(f"qnet_{i}", "grad"), g[f"s{s}_q{i}_grads"]) < 2e-3, (s, i)
            assert np.abs(a.get_params(f"qnet_{i}") - g[f"s{s}_q{i}_params"]).max() < 0.3 * kw["lr_critic"]
            assert rel(a.get_params(f"qnet_tgt_{i}"), g[f"s{s}_q{i}_tgt_params"]) < 1e-5
        assert np.abs(a.get_params("pi") - g[f"s{s}_pi_params"]).max() < 0.3 * kw["lr_actor"]
        assert abs(float(a.get_params("log_alpha")[0]) - g[f"s{s}_log_alpha"]) < 1e-6
    assert a.n_opts == steps
    if kw.get("adamw_critic", {}).get("amsgrad"):   # AdamW{amsgrad: true}: max_exp_avg_sq (parameter model +400) and exp_avg_sq after the last step
        for i in range(nc):
            v, vmax = a.get_params(f"qnet_{i}", "exp_avg_sq"), a.get_params(f"qnet_{i}", "max_exp_avg_sq")
            assert (vmax >= v).all() and (vmax > v * 1.1).mean() > 0.1          # the maximum really is ahead of the decayed second moment
            assert np.abs(vmax - g[f"q{i}_max_exp_avg_sq"]).max() <= 2e-3 * np.abs(g[f"q{i}_max_exp_avg_sq"]).max()
            assert np.abs(v - g[f"q{i}_exp_avg_sq"]).max() <= 2e-3 * np.abs(g[f"q{i}_exp_avg_sq"]).max()
        with pytest.raises(Exception):
            a.get_params("pi", "max_exp_avg_sq")                                   # the actor's AdamW has amsgrad off: no such arena
    a.close()


def test_sac_twin_q_auto_alpha(B, golden_dir):
    _run(B, "sac_17_6_twinq_auto", golden_dir)


def test_sac_twin_q_adamw_actor_and_amsgrad_critics(B, golden_dir):
    """OptimizerConfig::AdamW through bdr_sac_config::opt_actor / opt_critic (opt.rs:20-27, 38-55): decoupled decay, custom betas / eps
    on the actor; the twin critics with amsgrad.  Rewards are scaled 10x then 0.1x (make_golden.sac_case_batch), so the running
    maximum is ahead of exp_avg_sq and a plain-AdamW step would miss the golden parameters."""
    _run(B, "sac_17_6_twinq_adamw", golden_dir)


def test_sac_pendulum_fix_alpha_huber(B, golden_dir):
    _run(B, "sac_pendulum_fix_huber", golden_dir)


def test_sac_baseline_shape_b1024_vs_oracle(B):
    """BASELINE config 5: obs 17 / act 6, batch 1024, twin-Q [256,256], one update against the C oracle."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    od, ad, pu, qu, nc, Bsz = 17, 6, [256, 256], [256, 256], 2, 1024
    pi0 = T.init_params(T.sac_pi_shapes(od, pu, ad), 21) * np.float32(0.5)
    q0 = [T.init_params(T.sac_q_shapes(od, ad, qu), 22 + i) for i in range(nc)]
    kw = dict(lr_actor=3e-4, lr_critic=3e-4, ent_coef=("Auto", -6.0, 3e-4), critic_loss="Mse")
    a = _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0)
    ref = O.SacOracle(od, ad, pu, qu, pi0, q0, **kw)
    batch = T.sac_batch(Bsz, od, ad, 99)
    rec = a.update_on_batch(*batch)
    r = ref.update(*batch)
    # north_star's bar, on the Q-values: every critic evaluation of the update to 1e-4 of the largest |Q| (qvals, sac/base.rs:89-105) -
    # Q_i(obs, a_pi) of update_actor, Q_i(obs, act) of update_critic, the target critics on (next_obs, a') and their minimum,
    # and the TD target built from them
    QTOL = 1e-4
    for k in ("q_pi", "q_pred", "q_next"):
        assert rel(a.probe(k, Bsz), r[k]) < QTOL, (k, rel(a.probe(k, Bsz), r[k]))
    assert rel(a.probe("qvals_min", Bsz), r["q_next"].min(axis=0)) < QTOL
    assert rel(a.probe("tgt", Bsz), r["tgt"]) < QTOL, rel(a.probe("tgt", Bsz), r["tgt"])
    assert rel(a.probe("next_act", Bsz), r["next_a"]) < QTOL
    # log p = sum(-z^2/2 - ln sqrt(2 pi)) - sum ln(1 - a^2 + eps): ill-conditioned where |a| -> 1 (1 - a^2 cancels; eps = 1e-4), so the
    # rows are held to 1e-4 of the largest |log p| except the saturated ones (|a| > 0.995 in some dimension), which get the
    # conditioning's factor
    for k, acts in (("log_p", r["a"]), ("next_log_p", r["next_a"])):
        got, want = a.probe(k, Bsz), r[k]
        sat = (np.abs(acts) > 0.995).any(axis=1)
        scale = np.abs(want).max()
        assert np.abs(got - want)[~sat].max() < QTOL * scale, (k, np.abs(got - want)[~sat].max() / scale)
        assert np.abs(got - want).max() < 2e-3 * scale, (k, np.abs(got - want).max() / scale)
    # the critic loss is a mean over Q-values: 1e-4; the actor loss contains mean(alpha * log p): the saturated rows' conditioning
    assert abs(rec["loss_critic"] - r["loss_critic"]) <= QTOL * abs(r["loss_critic"]) + 1e-6, (rec["loss_critic"], r["loss_critic"])
    for k in ("loss_actor", "ent_coef"):
        assert abs(rec[k] - r[k]) <= 5e-4 * abs(r[k]) + 1e-6, (k, rec[k], r[k])
    assert rel(a.get_params("pi", "grad"), r["pi_grads"]) < 2e-3
    for i in range(nc):
        assert rel(a.get_params(f"qnet_{i}", "grad"), r["q_grads"][i]) < 2e-3
    a.close()


@pytest.mark.parametrize("od,ad,pu,qu,nc,Bsz,ent", [
    (11, 3, [96, 40], [72, 136], 3, 72, ("Auto", -3.0, 1e-3)),      # nothing a multiple of 32 / 64: every padding and row guard
    (5, 2, [64], [300], 1, 200, ("Fix", 0.2)),                        # one trunk layer, one critic, a 320-wide (padded) layer
    (23, 7, [128, 64, 32], [64, 64, 64], 4, 33, ("Auto", -7.0, 3e-4)),  # three trunk layers, four critics (12 dW GEMMs in one group), 33 rows
])
def test_sac_ragged_shapes_vs_oracle(B, od, ad, pu, qu, nc, Bsz, ent):
    """The latency-shaped SAC kernels (32x32 split-reduction tiles, grouped dW launch, fused reduce+Adam+track over several
    networks) on shapes that are not multiples of any tile: two updates against the C oracle - losses, every gradient, every
    parameter, the targets and log_alpha."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    pi0 = T.init_params(T.sac_pi_shapes(od, pu, ad), 31) * np.float32(0.5)
    q0 = [T.init_params(T.sac_q_shapes(od, ad, qu), 40 + i) for i in range(nc)]
    kw = dict(lr_actor=1e-3, lr_critic=2e-3, ent_coef=ent, critic_loss="SmoothL1")
    a = _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0)
    ref = O.SacOracle(od, ad, pu, qu, pi0, q0, **kw)
    for s in range(2):
        batch = T.sac_batch(Bsz, od, ad, 500 + s)
        rec = a.update_on_batch(*batch)
        r = ref.update(*batch)
        for k in ("loss_critic", "loss_actor", "ent_coef"):
            assert abs(rec[k] - r[k]) <= 5e-4 * abs(r[k]) + 1e-6, (s, k, rec[k], r[k])
        assert rel(a.get_params("pi", "grad"), r["pi_grads"]) < 2e-3, s
        for i in range(nc):
            assert rel(a.get_params(f"qnet_{i}", "grad"), r["q_grads"][i]) < 2e-3, (s, i)
    assert np.abs(a.get_params("pi") - ref.pi).max() < 0.3 * kw["lr_actor"]
    for i in range(nc):
        assert np.abs(a.get_params(f"qnet_{i}") - ref.qs[i]).max() < 0.3 * kw["lr_critic"]
        assert rel(a.get_params(f"qnet_tgt_{i}"), ref.qs_tgt[i]) < 1e-4
    a.close()


@pytest.mark.parametrize("od,ad,pu,qu,nc,Bsz,ent,loss", [
    (17, 6, [256, 256], [256, 256], 2, 1024, ("Auto", -6.0, 3e-4), "Mse"),          # BASELINE config 5
    (11, 3, [96, 40], [72, 136], 3, 72, ("Auto", -3.0, 1e-3), "SmoothL1"),           # three critics (an odd pair count), ragged rows
    (5, 2, [64], [300], 1, 200, ("Fix", 0.2), "Mse"),                                # two-layer critic: the last layer's dX feeds the first layer directly
    (23, 7, [128, 64, 32], [64, 64, 64], 4, 33, ("Auto", -7.0, 3e-4), "SmoothL1"),   # four critics (8 pairs), 33 rows, action columns 23..29 of the tile
    (30, 8, [64, 64], [64, 64], 2, 64, ("Fix", 1.0), "Mse"),                         # obs % 32 + act > 32: not covered by the row-block kernels (same path twice)
])
def test_sac_row_block_kernels_equal_the_layer_by_layer_path(B, monkeypatch, od, ad, pu, qu, nc, Bsz, ent, loss):
    """sac_fused.hpp: heads + action, critics' last layer + selection + EntCoef::update, d qmin / d a + tanh-Gaussian backward + heads'
    dX, target critics' last layer + TD target + losses are row-block kernels built on the layer-by-layer path's tile function;
    batch-wide sums are done by the last workgroup in the single-workgroup kernels' order.  Three updates on fixed minibatches:
    every loss, every parameter, gradient, Adam moment, target network, log_alpha and every probe identical bit for bit."""
    from oracle import torch_ref as T
    pi0 = T.init_params(T.sac_pi_shapes(od, pu, ad), 31) * np.float32(0.5)
    q0 = [T.init_params(T.sac_q_shapes(od, ad, qu), 40 + i) for i in range(nc)]
    kw = dict(lr_actor=1e-3, lr_critic=2e-3, ent_coef=ent, critic_loss=loss)
    outs = []
    for fuse in (True, False):
        if fuse: monkeypatch.delenv("BDR_NO_SAC_FUSE", raising=False)
        else: monkeypatch.setenv("BDR_NO_SAC_FUSE", "1")
        a = _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0)
        recs = [a.update_on_batch(*T.sac_batch(Bsz, od, ad, 700 + s)) for s in range(3)]
        names = ["pi", "log_alpha"] + [f"qnet_{i}" for i in range(nc)] + [f"qnet_tgt_{i}" for i in range(nc)]
        out = {n: a.get_params(n) for n in names}
        out.update({n + "/grad": a.get_params(n, "grad") for n in names if n != "log_alpha" and not n.startswith("qnet_tgt")})
        out.update({n + "/exp_avg_sq": a.get_params(n, "exp_avg_sq") for n in ("pi", "qnet_0")})
        out.update({k: a.probe(k, Bsz) for k in ("q_pred", "q_next", "qvals_min", "next_log_p", "tgt", "q_pi", "log_p", "next_act")})
        outs.append((out, recs))
        a.close()
    (f, frec), (u, urec) = outs
    assert frec == urec, (frec, urec)
    for k in f:
        assert (f[k] == u[k]).all(), (k, np.abs(f[k].astype(np.float64) - u[k]).max())


@pytest.mark.parametrize("od,ad,pu,qu,nc,Bsz,ent", [
    (17, 6, [256, 256], [256, 256], 2, 1024, ("Auto", -6.0, 3e-4)),   # BASELINE config 5 (64 -> 256 -> 256)
    (17, 6, [256, 256], [256, 256], 2, 1000, ("Auto", -6.0, 3e-4)),   # a last row block of 8 rows
    (11, 3, [256, 40], [256, 136], 3, 72, ("Auto", -3.0, 1e-3)),      # second layers of 64 / 192 (padded) columns: one tile per workgroup only; three critics (4 + 2 passes, then 3)
    (40, 6, [256, 64], [256, 128], 1, 96, ("Fix", 0.2)),               # one critic (2 passes, then 1), a 128-wide second layer
])
def test_sac_two_layer_chain_kernel_equals_two_launches(B, monkeypatch, od, ad, pu, qu, nc, Bsz, ent):
    """dense_chain.hpp: the two wide layers of the actor's trunk and of every critic pass run in ONE launch (32 rows per workgroup, the
    first layer's output kept in LDS).  Its tiles use the k-slices, MFMA order and four-way sum of the layer-by-layer kernel, in both of
    its forms (a wave per tile / a wave per k-slice): three updates, every parameter, gradient, moment, probe and loss bit for bit equal
    to BDR_NO_SAC_CHAIN=1.  Likewise the batch-wide parts of k_sac_q_last / k_sac_td_last (EntCoef::update, the loss sums), which run as one
    more workgroup of the launch behind them (k_dense_small_dx_tail) unless BDR_SAC_TAIL_IN_KERNEL=1 keeps them with their last workgroup, and
    the heads + action + log-probability part done by the last workgroup of each row block of the actor's trunk launch (k_sac_pi_chain_heads,
    BDR_SAC_HEADS_FUSE=1: same bits, slower - not the default)."""
    from oracle import torch_ref as T
    pi0 = T.init_params(T.sac_pi_shapes(od, pu, ad), 31) * np.float32(0.5)
    q0 = [T.init_params(T.sac_q_shapes(od, ad, qu), 40 + i) for i in range(nc)]
    kw = dict(lr_actor=1e-3, lr_critic=2e-3, ent_coef=ent, critic_loss="Mse")
    outs = []
    for mode in ("off", "auto", "1", "4", "tail", "heads"):
        for k in ("BDR_NO_SAC_CHAIN", "BDR_SAC_CHAIN_TPW", "BDR_SAC_TAIL_IN_KERNEL", "BDR_SAC_HEADS_FUSE"): monkeypatch.delenv(k, raising=False)
        if mode == "off": monkeypatch.setenv("BDR_NO_SAC_CHAIN", "1"); monkeypatch.setenv("BDR_SAC_TAIL_IN_KERNEL", "1")
        elif mode == "tail": monkeypatch.setenv("BDR_SAC_TAIL_IN_KERNEL", "1")
        elif mode == "heads": monkeypatch.setenv("BDR_SAC_HEADS_FUSE", "1")
        elif mode != "auto": monkeypatch.setenv("BDR_SAC_CHAIN_TPW", mode)
        a = _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0)
        recs = [a.update_on_batch(*T.sac_batch(Bsz, od, ad, 900 + s)) for s in range(3)]
        names = ["pi", "log_alpha"] + [f"qnet_{i}" for i in range(nc)] + [f"qnet_tgt_{i}" for i in range(nc)]
        out = {n: a.get_params(n) for n in names}
        out.update({n + "/grad": a.get_params(n, "grad") for n in names if n != "log_alpha" and not n.startswith("qnet_tgt")})
        out.update({n + "/exp_avg_sq": a.get_params(n, "exp_avg_sq") for n in ("pi", "qnet_0")})
        out.update({k: a.probe(k, Bsz) for k in ("q_pred", "q_next", "qvals_min", "next_log_p", "tgt", "q_pi", "log_p", "next_act")})
        outs.append((mode, out, recs))
        a.close()
    _, ref, rrec = outs[0]
    for mode, out, recs in outs[1:]:
        assert recs == rrec, (mode, recs, rrec)
        for k in ref:
            assert (out[k] == ref[k]).all(), (mode, k, np.abs(out[k].astype(np.float64) - ref[k]).max())


def test_sac_opt_over_replay_and_sample(B, tmp_path):
    """Agent::opt over the HBM ring with device-generated noise: finite losses, counters, checkpoint round trip,
    Policy::sample in eval mode == tanh(mean) of the oracle actor."""
    from oracle import torch_ref as T
    import torch
    rng = np.random.default_rng(3)
    od, ad = 17, 6
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=5000, seed=42), (od,), np.float32, (ad,), np.float32)
    n = 2000
    rb.push(rng.standard_normal((n, od)).astype(np.float32), rng.uniform(-1, 1, (n, ad)).astype(np.float32),
            rng.standard_normal((n, od)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
            (rng.random(n) < .05).astype(np.int8), np.zeros(n, np.int8))
    cfg = B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=128,
                      ent_coef_mode=("Auto", -6.0, 3e-4), n_updates_per_opt=2, device=0, seed=5)
    a = B.Sac.build(cfg)
    a.train()
    for _ in range(3):
        rec = a.opt_with_record(rb)
        assert all(np.isfinite(v) for v in rec.values()), rec
    assert a.n_opts == 6
    a.eval()
    obs = rng.standard_normal((9, od)).astype(np.float32)
    pi = a.get_params("pi")
    sac = T.TorchSac(od, ad, [64, 64], [64, 64], pi, [a.get_params("qnet_0"), a.get_params("qnet_1")], lr_actor=0, lr_critic=0)
    mean, _ = sac.pi_forward(torch.from_numpy(obs))
    assert rel(a.sample(obs), mean.tanh().detach().numpy()) < 1e-4
    files = a.save_params(str(tmp_path))
    assert all(os.path.exists(f) for f in files)
    # sac/base.rs:313-334: file names and order; ent_coef holds `log_alpha` (sac/ent_coef.rs)
    assert [os.path.basename(f) for f in files] == ["qnet_0.pt.tch", "qnet_tgt_0.pt.tch", "qnet_1.pt.tch", "qnet_tgt_1.pt.tch",
                                                    "pi.pt.tch", "ent_coef.pt.tch"]
    assert [n for n, _ in torch.jit.load(files[-1]).named_parameters()] == ["log_alpha"]
    b = B.Sac.build(cfg)
    b.load_params(str(tmp_path))
    assert (b.get_params("pi") == pi).all() and (b.get_params("qnet_tgt_1") == a.get_params("qnet_tgt_1")).all()
    a.close(); b.close(); rb.close()


def test_sac_opt_from_captured_graph_is_bit_identical_to_eager_launches(B, monkeypatch):
    """opt() replays its ~70 launches from a hipGraph whose varying arguments (Adam bias corrections, noise counter, replay
    stream position) are patched per step (csrc/step_graph.hpp).  Same seeds, same pushes: every parameter, the target nets,
    log_alpha and the recorded losses equal the eager path bit for bit - across pushes between opts (the ring grows), a mid-run
    update_on_batch with a LARGER batch (buffers re-allocated: the graph is re-captured) and n_updates_per_opt = 2.  The same for
    the sample drawn inside the pack kernel (replay_sample_plan) against the separate gather launch, incl. the buffer's stream
    position afterwards."""
    from oracle import torch_ref as T
    od, ad = 17, 6
    def run(env):
        for k in ("BDR_NO_STEP_GRAPH", "BDR_NO_STEP_GATHER", "BDR_NO_SMALL_GEMM", "BDR_STEP_GRAPH", "BDR_STEP_GRAPH_BREAK_AT", "BDR_NO_SAC_FUSE", "BDR_SAC_SIDE_QUEUE"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            if k == "BREAK": monkeypatch.setenv("BDR_STEP_GRAPH_BREAK_AT", "2")   # the third replay pass "diverges"
            elif k == "ONE_QUEUE": monkeypatch.setenv("BDR_SAC_SIDE_QUEUE", "0")
            elif k != "ADAPTIVE": monkeypatch.setenv(k, "1")
        if "BDR_NO_STEP_GRAPH" not in env and "ADAPTIVE" not in env: monkeypatch.setenv("BDR_STEP_GRAPH", "1")   # every opt from the graph
        rng = np.random.default_rng(11)
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=3000, seed=7), (od,), np.float32, (ad,), np.float32)
        def push(n):
            rb.push(rng.standard_normal((n, od)).astype(np.float32), rng.uniform(-1, 1, (n, ad)).astype(np.float32),
                    rng.standard_normal((n, od)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
                    (rng.random(n) < .05).astype(np.int8), np.zeros(n, np.int8))
        push(1000)
        cfg = B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=128,
                          ent_coef_mode=("Auto", -6.0, 3e-4), n_updates_per_opt=2, device=0, seed=5)
        a = B.Sac.build(cfg)
        a.train()
        import ctypes as C
        L = B._lib.lib()
        uid = (C.c_uint8 * B._lib.BDR_UNIQUE_ID_BYTES)()
        B._lib.check(L.bdr_comm_get_unique_id(uid))
        comm = C.c_void_p()
        B._lib.check(L.bdr_comm_init_rank(uid, 1, 0, 0, C.byref(comm)))
        recs, acts = [], []
        for k in range(12):
            recs.append(a.opt_with_record(rb) if k % 3 == 0 else (a.opt(rb), None)[1])
            if k % 2 == 1: push(100)
            if k == 6: a.update_on_batch(*T.sac_batch(256, od, ad, 99))
            # work on the agent's stream between opts that the two-queue sequence has to order itself behind: the in-stream parameter
            # exchange (identity at one rank) and Policy::sample (its own rows through the current buffer set, noise from the agent's stream)
            if k in (4, 9): B._lib.check(L.bdr_agent_allreduce_params(a.handle, comm, 0))
            if k in (2, 5, 10): acts.append(a.sample(rng.standard_normal((3, od)).astype(np.float32)))
        B._lib.check(L.bdr_comm_destroy(comm))
        out = {n: a.get_params(n) for n in ("pi", "qnet_0", "qnet_1", "qnet_tgt_0", "qnet_tgt_1", "log_alpha")}
        out["next_indices"] = rb.sample_indices(50)
        out["sampled_actions"] = np.concatenate(acts)
        n_opts = a.n_opts
        a.close(); rb.close()
        return out, recs, n_opts
    g, grec, gn = run(())
    assert gn == 25 and np.isfinite(g["pi"]).all()
    # eager launches; the separate gather launch instead of the sample drawn inside the pack kernel; both
    # ... and the default policy, which switches between graph and eager launches by whether the stream is idle when opt() is entered
    # ... and a replay pass that finds the sequence changed (forced): host counters are restored, THAT step is enqueued eagerly -
    # not dropped, Adam / RNG / replay positions not skewed - and the agent stays eager (step_graph_run)
    # ... and the row-block kernels that fuse the narrow layers into their neighbours (sac_fused.hpp: 21 launches) against the
    # layer-by-layer sequence (30 launches), from the graph and eagerly
    # ... and the two-queue sequence (csrc/sac.hip `side`: the next update's sample + actor forward beside this update's critic phase, in
    # the other buffer set, ordered by device flags; what an eager opt() takes unless BDR_SAC_SIDE_QUEUE=0) against one queue
    for env in (("BDR_NO_STEP_GRAPH",), ("BDR_NO_STEP_GRAPH", "ONE_QUEUE"), ("BDR_NO_STEP_GATHER",), ("BDR_NO_STEP_GRAPH", "BDR_NO_STEP_GATHER"),
                ("ADAPTIVE",), ("ADAPTIVE", "ONE_QUEUE"), ("BREAK",), ("BDR_NO_SAC_FUSE",), ("BDR_NO_SAC_FUSE", "BDR_NO_STEP_GRAPH"),
                ("BDR_NO_SAC_FUSE", "BDR_NO_STEP_GRAPH", "ONE_QUEUE")):
        e, erec, en = run(env)
        assert en == gn, env
        for k in g: assert (g[k] == e[k]).all(), (env, k)
        assert grec == erec, env


def test_device_noise_stream_moments_reproducibility_and_disjointness(B):
    """The N(0,1) draws of action_logp come from a counter-based device generator (sac.hip k_randn; the reference uses
    torch's global CPU generator, sac/base.rs:76, so only the distribution can be pinned): moments of a large draw, the same
    (seed, counter) gives the same numbers, consecutive draws continue the stream without overlap, another seed decorrelates."""
    def agent(seed):
        return B.Sac.build(B.SacConfig(obs_dim=3, act_dim=1, pi_units=(64,), q_units=(64,), batch_size=8, device=0, seed=seed))
    a, b, c = agent(11), agent(11), agent(12)
    n = 1 << 20
    x = a.draw_noise(n).astype(np.float64)
    assert np.isfinite(x).all()
    assert abs(x.mean()) < 4 / np.sqrt(n) and abs(x.var() - 1.0) < 0.01
    assert abs((x ** 3).mean()) < 0.02 and abs((x ** 4).mean() - 3.0) < 0.05          # skewness 0, kurtosis 3
    assert abs(np.mean(np.abs(x) < 1.0) - 0.682689) < 0.003 and abs(np.mean(np.abs(x) > 3.0) - 0.0027) < 0.0005
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.005                              # neighbouring counters are independent
    y1, y2 = b.draw_noise(1000), b.draw_noise(n - 1000)
    assert (np.concatenate([y1, y2]) == x.astype(np.float32)).all()                   # same stream, split anywhere
    z = c.draw_noise(n).astype(np.float64)
    assert abs(np.corrcoef(x, z)[0, 1]) < 0.005 and not (z[:64] == x[:64]).any()
    # an update consumes 2 * B * act_dim draws: the stream position after opt() is where a fresh agent is after as many draws
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100, seed=1), (3,), np.float32, (1,), np.float32)
    rb.fill_synthetic(100, seed=1, kind=1, n_actions=0)
    d, e = agent(5), agent(5)
    d.opt(rb); d.sync()
    e.draw_noise(2 * 8 * 1)
    assert (d.draw_noise(256) == e.draw_noise(256)).all()
    for h in (a, b, c, d, e):
        h.close()
    rb.close()


@pytest.mark.gpu
def test_sac_flag_wait_timeout_is_reported_and_the_agent_continues_on_one_queue(B):
    """The two-queue SAC step orders its queues with device flags (csrc/queue_flags.hpp).  Forced failure (BDR_SAC_STALL_AT=3): the side
    queue does not publish the third update's prologue, so the main queue's wait for it must time out (20 ms limit).  Required: no hang;
    the error word poisons the agent - later waits return at once and every kernel that writes parameters, moments, targets or the
    entropy coefficient skips its update, so the state stays that of the last good update (bit for bit the state of an undisturbed run
    after two updates) however the unordered kernels behind the failed wait interleave; the next synchronisation reports the failure
    as a gate error and clears it; the agent says it continues on one queue and trains on."""
    import subprocess
    import sys
    import tempfile
    script = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import border_amd as B
od, ad = 17, 6
rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=3000, seed=7), (od,), np.float32, (ad,), np.float32)
rb.fill_synthetic(3000, seed=1, kind=1, n_actions=0)
a = B.Sac.build(B.SacConfig(obs_dim=od, act_dim=ad, pi_units=(64, 64), q_units=(64, 64), n_critics=2, batch_size=128,
                            ent_coef_mode=("Auto", -6.0, 3e-4), device=0, seed=5))
a.train()
names = ("pi", "qnet_0", "qnet_1", "qnet_tgt_0", "qnet_tgt_1", "log_alpha")
state = lambda: np.concatenate([a.get_params(n).ravel() for n in names] + [a.get_params("pi", "exp_avg").ravel(), a.get_params("qnet_1", "exp_avg_sq").ravel()])
n_first = int(sys.argv[2])
for _ in range(n_first): a.opt(rb)
try:
    a.sync(); print("NO_ERROR")
except B.BdrError as e:
    print("ERR", e.code, "gate" in str(e) and "sac" in str(e))
a.sync()
s0 = state()
np.save(sys.argv[1], s0)
# the host's step counters are those of the state on the device (rolled back with the skipped updates): the same update on the same
# batch and noise gives the same bits as in the undisturbed run - Adam's bias corrections depend on the step numbers - and the agent's
# own noise stream continues from the same position
print("NOPTS", a.n_opts)
g = np.random.default_rng(11); Bn = 128
f32 = lambda x: x.astype(np.float32)
a.update_on_batch(f32(g.standard_normal((Bn, od))), f32(g.uniform(-1, 1, (Bn, ad))), f32(g.standard_normal((Bn, od))), f32(g.standard_normal(Bn)),
                  (g.random(Bn) < 0.1).astype(np.int8), f32(g.standard_normal((Bn, ad))), f32(g.standard_normal((Bn, ad))))
a.sync()
np.save(sys.argv[1] + ".s1.npy", np.concatenate([state(), a.draw_noise(256)]))
for _ in range(6): a.opt(rb)
rec = a.opt_with_record(rb)
print("LOSS", bool(np.isfinite(rec["loss_critic"])), "MOVED", bool((state() != s0).any()))
""" % os.path.join(os.path.dirname(__file__), "..")
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        for name, extra, n_first in (("stalled", {"BDR_SAC_STALL_AT": "3", "BDR_GATE_LIMIT_MS": "20"}, 7), ("clean", {}, 2)):
            env = dict(os.environ)
            for k in ("BDR_SAC_SIDE_QUEUE", "BDR_STEP_GRAPH", "BDR_NO_STEP_GRAPH", "BDR_SAC_STALL_AT"): env.pop(k, None)
            env.update(extra)
            path = os.path.join(d, name + ".npy")
            r = subprocess.run([sys.executable, "-c", script, path, str(n_first)], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-1500:]
            outs[name] = (r.stdout.split("\n"), r.stderr, np.load(path), np.load(path + ".s1.npy"))
    so, se, sp, sp1 = outs["stalled"]
    co, ce, cp, cp1 = outs["clean"]
    assert co[0] == "NO_ERROR" and co[1] == "NOPTS 2" and co[2] == "LOSS True MOVED True", co
    assert so[0].startswith("ERR") and " 3 " in so[0] + " " and so[0].endswith("True"), (so, se[-500:])
    assert so[2] == "LOSS True MOVED True", so
    assert "continues on one queue" in se, se[-500:]
    assert (sp == cp).all()          # seven updates enqueued, the third one's wait failed: the state is the one after two
    # ... and so are the host's counters (Sac::on_gate_timeout): n_opts, the three Adam step numbers (the next update's bits), the noise position
    assert so[1] == "NOPTS 2" and "were rolled back" in se, (so, se[-500:])
    assert (sp1 == cp1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("od,ad,pu,qu,nc,Bsz,ent,n_upd", [
    (11, 3, [96, 40], [72, 136], 3, 72, ("Auto", -3.0, 1e-3), 1),        # three critics, nothing a multiple of a tile
    (5, 2, [64], [300], 1, 200, ("Fix", 0.2), 3),                          # one trunk layer, one critic, three updates per opt
    (23, 7, [128, 64, 32], [64, 64, 64], 4, 33, ("Auto", -7.0, 3e-4), 2),  # three trunk layers, four critics, 33 rows
    (30, 8, [64, 64], [64, 64], 2, 64, ("Fix", 1.0), 1),                   # a shape the row-block kernels do not cover (layer-by-layer launches on two queues)
    (17, 6, [256, 256], [256, 256], 2, 256, ("Auto", -6.0, 3e-4), 2),      # BASELINE config 5's networks: the two-layer launches, the main queue's wait for the prologue INSIDE the first critic launch
])
def test_sac_two_queue_sequence_equals_one_queue_on_ragged_shapes(B, monkeypatch, od, ad, pu, qu, nc, Bsz, ent, n_upd):
    """Agent::opt over the ring, eager launches: the two-queue sequence (next update's sample + actor forward on the side queue, two
    buffer sets, device flags) against everything on one queue - pushes between opts, records, every parameter, target, Adam moment and
    the buffer's next indices bit for bit."""
    def run(side):
        for k in ("BDR_STEP_GRAPH", "BDR_NO_STEP_GRAPH", "BDR_SAC_SIDE_QUEUE"): monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("BDR_NO_STEP_GRAPH", "1")
        if not side: monkeypatch.setenv("BDR_SAC_SIDE_QUEUE", "0")
        rng = np.random.default_rng(23)
        rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=700, seed=9), (od,), np.float32, (ad,), np.float32)
        def push(n):
            rb.push(rng.standard_normal((n, od)).astype(np.float32), rng.uniform(-1, 1, (n, ad)).astype(np.float32),
                    rng.standard_normal((n, od)).astype(np.float32), rng.standard_normal(n).astype(np.float32),
                    (rng.random(n) < .05).astype(np.int8), np.zeros(n, np.int8))
        push(500)
        a = B.Sac.build(B.SacConfig(obs_dim=od, act_dim=ad, pi_units=tuple(pu), q_units=tuple(qu), n_critics=nc, batch_size=Bsz,
                                    ent_coef_mode=ent, n_updates_per_opt=n_upd, critic_loss="SmoothL1", device=0, seed=3))
        a.train()
        recs = []
        for k in range(60 if pu == [256, 256] else 9):   # (the in-kernel wait of the config-5 shape: 120 updates)
            recs.append(a.opt_with_record(rb) if k % 4 == 3 else (a.opt(rb), None)[1])
            if k % 2 == 0: push(37)        # the ring wraps (capacity 700) while updates are in flight
        names = ["pi", "log_alpha"] + [f"qnet_{i}" for i in range(nc)] + [f"qnet_tgt_{i}" for i in range(nc)]
        out = {n: a.get_params(n) for n in names}
        out["pi/exp_avg_sq"] = a.get_params("pi", "exp_avg_sq")
        out["next_indices"] = rb.sample_indices(40)
        a.close(); rb.close()
        return out, recs
    two, trec = run(True)
    one, orec = run(False)
    assert trec == orec
    for k in two:
        assert (two[k] == one[k]).all(), k
    assert np.isfinite(two["pi"]).all()
