"""C oracle (border_oracle.c) vs the committed PyTorch-CPU goldens (tests/golden/dqn_*.npz):
Q-values, TD target, loss, gradients, parameters after k Adam steps, target-net tracking.
Tolerances: the goldens are f32 ATen results (summation-order noise ~1e-6 rel); the bar for the
HIP path is 1e-4 rel on Q-values (BASELINE.json), the oracle itself is held to 2e-5."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import torch_ref as T


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _run(fix, net, shapes, batch_fn, n_steps, param_seed, **kw):
    g = np.load(fix)
    agent = O.DqnOracle(net, T.init_params(shapes, param_seed), **kw)
    st = None
    for s in range(n_steps):
        obs, act, nobs, rew, term = batch_fn(s)
        r = agent.update(obs, act, nobs, rew, term, probe=True)
        st = st or (max(1, r["grads"].size // 4096) | 1)
        assert rel(r["q_pred_all"], g[f"s{s}_q_pred_all"]) < 2e-5, s
        assert rel(r["q_next_all"], g[f"s{s}_q_next_all"]) < 2e-5, s
        assert rel(r["tgt"], g[f"s{s}_tgt"]) < 2e-5
        assert abs(r["loss"] - g[f"s{s}_loss"]) <= 2e-5 * abs(g[f"s{s}_loss"]) + 1e-9
        assert rel(r["grads"][::st], g[f"s{s}_grads_sample"]) < 5e-5, s
        o = 0
        for i, sh in enumerate(shapes):
            n = int(np.prod(sh))
            gn = np.linalg.norm(r["grads"][o:o + n].astype(np.float64))
            assert abs(gn - g[f"s{s}_grad_norms"][i]) <= 5e-5 * g[f"s{s}_grad_norms"][i] + 1e-12
            o += n
        # parameters: Adam's first steps move every weight by ~lr regardless of |g|, so compare the
        # *update* against lr rather than the parameter against itself
        d = np.abs(agent.q[::st].astype(np.float64) - g[f"s{s}_params_sample"])
        assert d.max() < 0.02 * kw["lr"], (s, d.max())
        assert rel(agent.q_tgt[::st], g[f"s{s}_tgt_params_sample"]) < 1e-5


def test_cnn_b4_huber(golden_dir):
    _run(os.path.join(golden_dir, "dqn_cnn_b4_huber.npz"), O.cnn_cfg(6), T.cnn_shapes(6),
         lambda s: T.synthetic_atari_batch(4, 6, 100 + s), 3, 1,
         lr=1e-4, critic_loss="SmoothL1", tau=1.0, soft_update_interval=2)


def test_cnn_b8_mse_double_dqn(golden_dir):
    _run(os.path.join(golden_dir, "dqn_cnn_b8_mse_ddqn.npz"), O.cnn_cfg(6), T.cnn_shapes(6),
         lambda s: T.synthetic_atari_batch(8, 6, 200 + s), 2, 2,
         lr=1e-4, critic_loss="Mse", double_dqn=True, tau=0.005, soft_update_interval=1)


def test_mlp_cartpole(golden_dir):
    def cart(s):
        rng = np.random.default_rng(300 + s)
        obs = rng.standard_normal((32, 4)).astype(np.float32)
        nobs = rng.standard_normal((32, 4)).astype(np.float32)
        act = rng.integers(0, 2, 32)
        return obs, act, nobs, np.ones(32, np.float32), (rng.random(32) < 0.1).astype(np.int8)

    _run(os.path.join(golden_dir, "dqn_mlp_cartpole.npz"), O.mlp_cfg(4, [64, 64], 2), T.mlp_shapes(4, [64, 64], 2),
         cart, 5, 3, lr=1e-3, critic_loss="Mse", tau=0.01, soft_update_interval=1)


def test_track_matches_candle_twin_kat():
    """border-candle-agent/src/util.rs:90-142 test_track: tau=0.7, src [1,2,3], dst [4,5,6]."""
    import ctypes as C
    src = np.array([1, 2, 3], np.float32)
    dst = np.array([4, 5, 6], np.float32)
    O.lib().orc_track(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), C.c_double(0.7), 3)
    exp = (np.float32(0.7) * src + np.float32(1.0 - 0.7) * np.array([4, 5, 6], np.float32))
    assert np.allclose(dst, exp, rtol=0, atol=0)


@pytest.mark.parametrize("adamw", [None, dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6), dict(beta1=0.8, beta2=0.9, wd=0.05, eps=1e-6, amsgrad=True)])
def test_restated_optimizer_step_equals_torch_optim(adamw):
    """TorchDqn._adam restates libtorch's Adam::step / AdamW::step by hand (the kernels mirror it element by element); here the
    restatement is held against torch.optim.Adam / torch.optim.AdamW(amsgrad=...) THEMSELVES on the same gradients - shrinking
    gradients, so that with amsgrad the running maximum of exp_avg_sq really differs from exp_avg_sq."""
    import torch
    shapes = T.mlp_shapes(4, [64, 64], 2)
    p0 = T.init_params(shapes, 23)
    lr = 2e-3
    t = T.TorchDqn("mlp", shapes, p0, lr=lr, critic_loss="SmoothL1", tau=0.01, soft_update_interval=1, adamw=adamw)
    ref = [x.detach().clone().requires_grad_(True) for x in T.unflatten(p0, shapes)]
    if adamw is None:
        opt = torch.optim.Adam(ref, lr=lr)
    else:
        opt = torch.optim.AdamW(ref, lr=lr, betas=(adamw["beta1"], adamw["beta2"]), eps=adamw["eps"], weight_decay=adamw["wd"],
                                amsgrad=bool(adamw.get("amsgrad", False)))
    g = torch.Generator().manual_seed(0)
    for step in range(6):
        scale = 10.0 if step < 2 else 0.1
        for p, r in zip(t.q, ref):
            grad = torch.randn(p.shape, generator=g) * scale
            p.grad = grad.clone()
            r.grad = grad.clone()
        t._adam()
        opt.step()
        for p, r in zip(t.q, ref):
            assert (p.detach() - r.detach()).abs().max().item() <= 2e-7 * max(1.0, r.detach().abs().max().item()), step
    if adamw is not None and adamw.get("amsgrad"):
        assert any((vm > v * 1.5).float().mean().item() > 0.2 for vm, v in zip(t.vmax, t.v))
