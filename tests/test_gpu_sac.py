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


def _agent(B, od, ad, pu, qu, nc, Bsz, kw, pi0, q0):
    cfg = B.SacConfig(obs_dim=od, act_dim=ad, pi_units=tuple(pu), q_units=tuple(qu), lr_actor=kw["lr_actor"], lr_critic=kw["lr_critic"],
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
        rec = a.update_on_batch(*T.sac_batch(Bsz, od, ad, seed + 100 + s))
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
    a.close()


def test_sac_twin_q_auto_alpha(B, golden_dir):
    _run(B, "sac_17_6_twinq_auto", golden_dir)


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
    for k in ("loss_critic", "loss_actor", "ent_coef"):
        assert abs(rec[k] - r[k]) <= 5e-4 * abs(r[k]) + 1e-6, (k, rec[k], r[k])
    assert rel(a.get_params("pi", "grad"), r["pi_grads"]) < 2e-3
    for i in range(nc):
        assert rel(a.get_params(f"qnet_{i}", "grad"), r["q_grads"][i]) < 2e-3
    a.close()


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
