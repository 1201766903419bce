"""C oracle IQN step (analytic gradients through the quantile-Huber loss, the Hadamard merge and the
cosine embedding) vs the committed PyTorch-autograd goldens."""
import os
import sys

import numpy as np

from oracle import oracle as O
from oracle import torch_ref as T

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _run(name, golden_dir):
    kind, F_, E, fu, A, pin, pu, B, n_p, n_t, steps, lr, sh, p0, seed = MG.iqn_case(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    agent = O.IqnOracle(kind, p0, lr=lr, feature_dim=F_, embed_dim=E, f_units=fu, n_actions=A, psi_in=pin or 0, psi_units=pu,
                        tau=0.01, soft_update_interval=2)
    st = MG.sample_stride(p0.size)
    for s in range(steps):
        r = agent.update(*T.iqn_batch(B, kind, A, n_p, n_t, seed + 50 + s, in_dim=pin))
        assert rel(r["z_pred"], g[f"s{s}_z_pred"]) < 2e-5 and rel(r["z_tgt"], g[f"s{s}_z_tgt"]) < 2e-5, s
        assert rel(r["tgt"], g[f"s{s}_tgt"]) < 2e-5
        assert abs(r["loss"] - g[f"s{s}_loss"]) <= 2e-5 * abs(g[f"s{s}_loss"]) + 1e-9
        assert rel(r["grads"][::st], g[f"s{s}_grads_sample"]) < 1e-4, (s, rel(r["grads"][::st], g[f"s{s}_grads_sample"]))
        assert np.abs(agent.p[::st].astype(np.float64) - g[f"s{s}_params_sample"]).max() < 0.05 * lr
        assert rel(agent.p_tgt[::st], g[f"s{s}_tgt_params_sample"]) < 1e-5


def test_iqn_mlp_small(golden_dir):
    _run("iqn_mlp_small", golden_dir)


def test_iqn_cnn_b2(golden_dir):
    _run("iqn_cnn_b2", golden_dir)


def test_cnn_feature_extractor_takes_other_frame_stack_depths():
    """AtariCnnConfig::n_stack on IQN's psi (cnn/config.rs:14-24): the restatement's parameter count and one tiny update per depth
    (psi_in carries n_stack for the cnn trunk; 0 = the default 4)."""
    for ns in (1, 2, 4, 8):
        sh = T.iqn_shapes("cnn", 3136, 64, [32], 3, n_stack=ns)
        p0 = T.init_params(sh[0] + sh[1] + sh[2], 5 + ns)
        assert sh[0][0] == (32, ns, 8, 8)
        ref = O.IqnOracle("cnn", p0, lr=1e-4, feature_dim=3136, embed_dim=64, f_units=[32], n_actions=3, psi_in=ns)
        batch = T.iqn_batch(2, "cnn", 3, 4, 4, 9 + ns, n_stack=ns)
        r = ref.update(*batch)
        assert np.isfinite(r["loss"]) and np.isfinite(r["grads"]).all()
        assert np.abs(r["grads"][:2048 * ns]).max() > 0          # conv1's weights receive a gradient
    assert O.IqnOracle("cnn", T.init_params(sum(T.iqn_shapes("cnn", 3136, 64, [32], 3), []), 1), lr=1e-4, feature_dim=3136, embed_dim=64,
                       f_units=[32], n_actions=3).p.size == sum(int(np.prod(s)) for s in sum(T.iqn_shapes("cnn", 3136, 64, [32], 3), []))
