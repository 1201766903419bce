"""HIP IQN step (through the C ABI) vs the committed PyTorch-autograd goldens and the C oracle.
BASELINE config 4 shape: AtariCnn{skip_linear} trunk, F = 3136, embed 64, merge Mlp(3136,[512],A)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402

QTOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def B():
    import border_amd
    if border_amd.device_count() == 0:
        pytest.fail("no MI355X visible: the HIP path must run on the GPU box")
    return border_amd


def _agent(B, kind, F_, E, fu, A, pin, pu, Bsz, lr, p0, n_stack=4, **kw):
    f_cfg = B.AtariCnnConfig(n_stack=n_stack, skip_linear=True) if kind == "cnn" else B.MlpConfig(in_dim=pin, units=tuple(pu), out_dim=F_, activation_out=True)
    cfg = B.IqnConfig(f_config=f_cfg, feature_dim=F_, embed_dim=E, m_units=tuple(fu), n_actions=A, lr=lr, batch_size=Bsz, device=0, **kw)
    a = B.Iqn.build(cfg)
    a.set_params(p0, "iqn"); a.set_params(p0, "iqn_tgt")
    return a


def _run(B, name, golden_dir):
    from oracle import torch_ref as T
    kind, F_, E, fu, A, pin, pu, Bsz, n_p, n_t, steps, lr, sh, p0, seed = MG.iqn_case(name)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    adamw = MG.IQN_ADAMW.get(name)
    opt = {} if adamw is None else dict(opt_config=B.OptimizerConfig.AdamW(lr, **adamw))
    a = _agent(B, kind, F_, E, fu, A, pin, pu, Bsz, lr, p0, tau=0.01, soft_update_interval=2, **opt)
    assert (a.get_params("iqn") == p0).all() and a.param_count() == p0.size
    st = MG.sample_stride(p0.size)
    for s in range(steps):
        batch = MG.iqn_case_batch(name, s)
        assert rel(a.forward(batch[0], batch[5], "iqn"), g[f"s{s}_z_pred"]) < QTOL, s
        assert rel(a.forward(batch[2], batch[6], "iqn_tgt"), g[f"s{s}_z_tgt"]) < QTOL, s
        rec = a.update_on_batch(*batch)
        assert abs(rec["loss_critic"] - g[f"s{s}_loss"]) <= QTOL * abs(g[f"s{s}_loss"]) + 1e-9, (s, rec, g[f"s{s}_loss"])
        grads = a.get_params("grad")
        assert rel(grads[::st], g[f"s{s}_grads_sample"]) < 5e-4, (s, rel(grads[::st], g[f"s{s}_grads_sample"]))
        o = 0
        for i, shp in enumerate(sh[0] + sh[1] + sh[2]):
            n = int(np.prod(shp))
            gn = np.linalg.norm(grads[o:o + n].astype(np.float64))
            assert abs(gn - g[f"s{s}_grad_norms"][i]) <= 1e-3 * g[f"s{s}_grad_norms"][i] + 1e-12, (s, i)
            o += n
        assert np.abs(a.get_params("iqn")[::st].astype(np.float64) - g[f"s{s}_params_sample"]).max() < 0.1 * lr
        assert rel(a.get_params("iqn_tgt")[::st], g[f"s{s}_tgt_params_sample"]) < 1e-5
    if adamw is not None and adamw.get("amsgrad"):   # max_exp_avg_sq (arena 5) after the last step
        v, vmax = a.get_params("exp_avg_sq"), a.get_params("max_exp_avg_sq")
        assert (vmax >= v).all() and (vmax > v * 1.1).mean() > 0.3
        assert np.abs(vmax - g["max_exp_avg_sq"]).max() <= 2e-3 * np.abs(g["max_exp_avg_sq"]).max()
        assert np.abs(v - g["exp_avg_sq"]).max() <= 2e-3 * np.abs(g["exp_avg_sq"]).max()
    a.close()


def test_iqn_golden_mlp_small(B, golden_dir):
    _run(B, "iqn_mlp_small", golden_dir)


def test_iqn_golden_mlp_small_adamw_amsgrad(B, golden_dir):
    """OptimizerConfig::AdamW{amsgrad: true} through bdr_iqn_config::opt (iqn/model/config.rs:50 -> opt.rs:20-27, 38-55), 4 steps with
    rewards scaled 10x then 0.1x so that the running maximum of exp_avg_sq is ahead of it (make_golden.iqn_case_batch)."""
    _run(B, "iqn_mlp_small_adamw", golden_dir)


def test_iqn_golden_cnn_b2(B, golden_dir):
    _run(B, "iqn_cnn_b2", golden_dir)


@pytest.mark.parametrize("A", [6, 4, 9, 18, 33])
def test_iqn_cnn_64_quantiles_vs_oracle(B, A):
    """BASELINE config 4 shape at a reduced batch (B=16, 64 pred/tgt quantiles, Nature trunk): one update vs the C oracle, at
    action counts on both sides of the merge head's 32- and 64-column padding (Atari: 3...18 actions, env.rs:97-103)."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    sh = T.iqn_shapes("cnn", 3136, 64, [512], A)
    p0 = T.init_params(sh[0] + sh[1] + sh[2], 31)
    a = _agent(B, "cnn", 3136, 64, [512], A, None, [], 16, 1e-4, p0, tau=1.0, soft_update_interval=10000)
    ref = O.IqnOracle("cnn", p0, lr=1e-4, feature_dim=3136, embed_dim=64, f_units=[512], n_actions=A, tau=1.0, soft_update_interval=10000)
    batch = T.iqn_batch(16, "cnn", A, 64, 64, 77)
    rec = a.update_on_batch(*batch)
    r = ref.update(*batch)
    assert abs(rec["loss_critic"] - r["loss"]) <= QTOL * abs(r["loss"])
    g = a.get_params("grad")
    assert rel(g, r["grads"]) < 5e-4, rel(g, r["grads"])
    a.close()


@pytest.mark.parametrize("ns", [1, 2, 8])
def test_iqn_other_frame_stack_depths_vs_oracle(B, ns, tmp_path):
    """AtariCnnConfig::n_stack (cnn/config.rs:14-24) on IQN's feature extractor, as on DQN's: conv1 has 64 * n_stack rows.  One update
    vs the C oracle, an opt over a ring of n_stack-frame rows, and the checkpoint's c1.weight shape."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    A, Bsz = 6, 8
    sh = T.iqn_shapes("cnn", 3136, 64, [512], A, n_stack=ns)
    p0 = T.init_params(sh[0] + sh[1] + sh[2], 33 + ns)
    a = _agent(B, "cnn", 3136, 64, [512], A, None, [], Bsz, 1e-4, p0, n_stack=ns, tau=1.0, soft_update_interval=10000)
    assert a.param_count() == p0.size and (a.get_params("iqn") == p0).all()
    ref = O.IqnOracle("cnn", p0, lr=1e-4, feature_dim=3136, embed_dim=64, f_units=[512], n_actions=A, psi_in=ns, tau=1.0, soft_update_interval=10000)
    batch = T.iqn_batch(Bsz, "cnn", A, 32, 32, 78 + ns, n_stack=ns)
    rec = a.update_on_batch(*batch)
    r = ref.update(*batch)
    assert abs(rec["loss_critic"] - r["loss"]) <= QTOL * abs(r["loss"])
    g = a.get_params("grad")
    assert rel(g, r["grads"]) < 5e-4, rel(g, r["grads"])
    n1 = 2048 * ns
    assert rel(g[:n1], r["grads"][:n1]) < 2e-3      # conv1's own weight gradient, on its own scale
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=100, seed=42), (ns, 1, 84, 84), np.uint8)
    rb.fill_synthetic(100, seed=3, kind=0, n_actions=A)
    for _ in range(2):
        a.opt(rb)
    a.sync()
    assert np.isfinite(a.get_params("iqn")).all()
    a.save_params(str(tmp_path / "m"))
    from border_amd import checkpoint as ck
    t = ck.read(str(tmp_path / "m" / "iqn.pt.tch"), [("c1.weight", (32, ns, 8, 8)), ("c1.bias", (32,))])     # shapes must match the file's
    assert (t["c1.weight"].ravel() == a.get_params("iqn")[:n1]).all()
    a.close(); rb.close()


def test_iqn_baseline_config4_full_size_vs_oracle(B):
    """BASELINE config 4 at its real size: batch 512, 64 prediction / 64 target quantiles, Nature-CNN trunk (F = 3136,
    embed 64, merge Mlp(3136, [512], 6)) - one Iqn::update_critic (iqn/base.rs:63-170) against the C oracle (OpenMP on the
    host cores: [32768][3136] x [3136][512] layers, seconds on the GPU box): quantile values of both networks <= 1e-4
    relative (north_star's bar on Q-values), the quantile-Huber loss, and every parameter gradient."""
    from oracle import oracle as O
    from oracle import torch_ref as T
    Bsz, NQ, A = 512, 64, 6
    sh = T.iqn_shapes("cnn", 3136, 64, [512], A)
    p0 = T.init_params(sh[0] + sh[1] + sh[2], 41)
    a = _agent(B, "cnn", 3136, 64, [512], A, None, [], Bsz, 1e-4, p0, tau=1.0, soft_update_interval=10000)
    ref = O.IqnOracle("cnn", p0, lr=1e-4, feature_dim=3136, embed_dim=64, f_units=[512], n_actions=A, tau=1.0, soft_update_interval=10000)
    batch = T.iqn_batch(Bsz, "cnn", A, NQ, NQ, 123)
    rec = a.update_on_batch(*batch)
    r = ref.update(*batch)
    assert abs(rec["loss_critic"] - r["loss"]) <= QTOL * abs(r["loss"]), (rec, r["loss"])
    g = a.get_params("grad")
    assert rel(g, r["grads"]) < 5e-4, rel(g, r["grads"])
    # Per variable, relative to that variable's own scale.  ATen itself (torch 2.10 CPU, f32) agrees with the oracle to
    # 1e-6 ... 7e-5 per variable on this very step.  The trunk has 512 x 21 120 ReLU units: a unit whose pre-activation is
    # within f32 round-off of zero may be masked differently by two correct implementations, which moves ONE output
    # channel of that conv layer's weight gradient (a single term of a 41 472-term sum, ~1/sqrt(41 472) of the sum) and,
    # more weakly, the layers below it - the same effect tests/test_gpu_dqn.py::assert_grads_close allows for.
    o = 0
    for shp in sh[0] + sh[1] + sh[2]:
        n = int(np.prod(shp))
        a_, b_ = g[o:o + n].astype(np.float64), r["grads"][o:o + n].astype(np.float64)
        d = np.abs(a_ - b_) / max(np.abs(b_).max(), 1e-30)
        bad = (d > 1e-3).reshape(shp[0], -1).any(1)
        assert bad.sum() <= 4 and d.max() < 2e-2, (shp, int(bad.sum()), d.max())
        o += n
    # parameters after the first Adam step: every element moves by lr * g / (|g| + 1e-8), i.e. by ~lr unless |g| is at the
    # 1e-8 level, where the step is as ill-conditioned as the f32 round-off of g itself - so: all but a sliver of the
    # 1.9 M elements within 10 % of lr, none further than the 2 lr two opposite steps can differ by
    p1 = a.get_params("iqn")
    dp = np.abs(p1.astype(np.float64) - ref.p)
    assert (dp > 0.1 * 1e-4).mean() < 2e-3 and dp.max() <= 2.01e-4, ((dp > 0.1 * 1e-4).mean(), dp.max())
    # forward values of the UPDATED online net and the target net on a 64-row slice (host copies of [64][64][6])
    sl = slice(0, 64)
    z_on = a.forward(batch[0][sl], batch[5][sl], "iqn")
    z_tg = a.forward(batch[2][sl], batch[6][sl], "iqn_tgt")
    assert rel(z_tg, r["z_tgt"][sl]) < QTOL, rel(z_tg, r["z_tgt"][sl])
    ref2 = O.IqnOracle("cnn", ref.p, lr=1e-4, feature_dim=3136, embed_dim=64, f_units=[512], n_actions=A)
    r2 = ref2.update(*[np.asarray(x)[sl] for x in batch])
    assert rel(z_on, r2["z_pred"]) < QTOL, rel(z_on, r2["z_pred"])
    a.close()


@pytest.mark.parametrize("Bsz,NQ", [(64, 64), (128, 32)])
def test_split_operand_merge_layer_equals_the_exact_kernels(B, monkeypatch, Bsz, NQ):
    """The merge layer f.L[1] at a matrix-bound size (M = B * N = 4096 rows x 3136 x 512) on the bf16 matrix cores with split operands
    (igemm_b3.hpp: six of the nine exact bf16 partial products) against the exact FP32-MFMA kernels (arithmetic = f32_exact) on the same
    update: quantile values 1e-5, loss 1e-5, every gradient 1e-4 of its variable's scale - an order tighter than the 1e-4 bar both
    hold against the oracle - and the profile labels name the arithmetic that ran."""
    from oracle import torch_ref as T
    A = 6
    sh = T.iqn_shapes("cnn", 3136, 64, [512], A)
    p0 = T.init_params(sh[0] + sh[1] + sh[2], 51)
    batch = T.iqn_batch(Bsz, "cnn", A, NQ, NQ, 321)
    out = {}
    for mode in ("exact", "split", "split_separate_merge_bwd"):
        monkeypatch.delenv("BDR_IQN_F32_EXACT", raising=False); monkeypatch.delenv("BDR_IQN_NO_MERGE_EPILOGUE", raising=False)
        if mode not in ("exact", "split"):      # with 64 percent points per sample the merge's backward is the input-gradient kernel's epilogue; this is the separate pass
            monkeypatch.setenv("BDR_IQN_NO_MERGE_EPILOGUE", "1")
        # the arithmetic is stated through the boundary (bdr_iqn_config::arithmetic), not through the A/B variable
        a = _agent(B, "cnn", 3136, 64, [512], A, None, [], Bsz, 1e-4, p0, tau=1.0, soft_update_interval=10000,
                   arithmetic="f32_exact" if mode == "exact" else "bf16x3_6")
        z = a.forward(batch[0], batch[5], "iqn")
        a.profile_enable(True)
        rec = a.update_on_batch(*batch)
        import bench
        labels = [l for l, _ in bench.read_profile(a)]
        a.profile_enable(False)
        out[mode] = (z, rec["loss_critic"], a.get_params("grad"), a.get_params("iqn"), labels)
        a.close()
    assert all(l in out["split"][4] for l in ("iqn_phi_3xbf16", "iqn_f_fwd1_3xbf16", "iqn_f_dx1_3xbf16", "iqn_f_dw1_3xbf16")) and "iqn_f_fwd1" in out["exact"][4]
    assert not any(l.endswith("3xbf16") for l in out["exact"][4])
    # (the merge's backward is the input-gradient kernel's epilogue when a wave's 64 rows are one sample's percent points)
    assert ("iqn_merge_bwd" not in out["split"][4]) == (NQ == 64) and "iqn_merge_bwd" in out["split_separate_merge_bwd"][4] and "iqn_merge_bwd" in out["exact"][4]
    g_f, g_p = out["split"][2].astype(np.float64), out["split_separate_merge_bwd"][2].astype(np.float64)
    assert np.abs(g_f - g_p).max() <= 2e-6 * np.abs(g_p).max()      # same products, the 64-row sums in another order
    assert rel(out["split"][0], out["exact"][0]) < 1e-5, rel(out["split"][0], out["exact"][0])
    assert abs(out["split"][1] - out["exact"][1]) <= 1e-5 * abs(out["exact"][1])
    o = 0
    for shp in sh[0] + sh[1] + sh[2]:
        n = int(np.prod(shp))
        g_s, g_e = out["split"][2][o:o + n].astype(np.float64), out["exact"][2][o:o + n].astype(np.float64)
        # (a hidden unit whose pre-activation is within the split's 4e-6 of zero may be masked differently by the two arithmetics: that moves
        # ONE output row of a layer's weight gradient by one sample's term - allowed for at most two rows per variable, as in the full-size test)
        d = np.abs(g_s - g_e) / max(np.abs(g_e).max(), 1e-30)
        bad = (d > 1e-4).reshape(shp[0], -1).any(1)
        assert bad.sum() <= 2 and d.max() < 5e-3, (shp, int(bad.sum()), d.max())
        o += n


@pytest.mark.parametrize("act_out", [True, False])
def test_split_operand_merge_layer_behind_an_mlp_feature_extractor(B, monkeypatch, act_out):
    """The same comparison with psi = Mlp(8 -> [64] -> 2048): the fused merge backward masks d psi by psi > 0 only when psi ends in a
    ReLU (MlpConfig::activation_out), and its feature rows have the Mlp's padded leading dimension."""
    from oracle import torch_ref as T
    Bsz, NQ, A, F_ = 64, 64, 5, 2048
    sh = T.iqn_shapes("mlp", F_, 64, [512], A, psi_in=8, psi_units=[64])
    p0 = T.init_params(sh[0] + sh[1] + sh[2], 57)
    batch = T.iqn_batch(Bsz, "mlp", A, NQ, NQ, 322, in_dim=8)
    out = {}
    for mode in ("exact", "split"):
        monkeypatch.delenv("BDR_IQN_F32_EXACT", raising=False)
        f_cfg = B.MlpConfig(in_dim=8, units=(64,), out_dim=F_, activation_out=act_out)
        cfg = B.IqnConfig(f_config=f_cfg, feature_dim=F_, embed_dim=64, m_units=(512,), n_actions=A, lr=1e-4, batch_size=Bsz, device=0, tau=1.0, soft_update_interval=10000,
                          arithmetic="f32_exact" if mode == "exact" else "bf16x3_6")
        a = B.Iqn.build(cfg)
        a.set_params(p0, "iqn"); a.set_params(p0, "iqn_tgt")
        a.profile_enable(True)
        rec = a.update_on_batch(*batch)
        import bench
        labels = [l for l, _ in bench.read_profile(a)]
        a.profile_enable(False)
        out[mode] = (rec["loss_critic"], a.get_params("grad"), labels)
        a.close()
    assert "iqn_f_dx1_3xbf16" in out["split"][2] and "iqn_merge_bwd" not in out["split"][2] and "iqn_merge_bwd" in out["exact"][2]
    assert abs(out["split"][0] - out["exact"][0]) <= 1e-5 * abs(out["exact"][0])
    o = 0
    for shp in sh[0] + sh[1] + sh[2]:
        n = int(np.prod(shp))
        g_s, g_e = out["split"][1][o:o + n].astype(np.float64), out["exact"][1][o:o + n].astype(np.float64)
        d = np.abs(g_s - g_e) / max(np.abs(g_e).max(), 1e-30)
        bad = (d > 1e-4).reshape(shp[0], -1).any(1)
        assert bad.sum() <= 2 and d.max() < 5e-3, (shp, int(bad.sum()), d.max())
        o += n


def test_iqn_opt_over_replay_and_qvalues(B, tmp_path):
    """Agent::opt over the HBM ring with device-drawn percent points (Uniform64, batch 32): finite loss, counters,
    checkpoint round trip; Policy::sample's averaged action values == mean over Const32's 33 points of forward()."""
    cap, Bsz, A = 2000, 32, 6
    rb = B.SimpleReplayBuffer(B.SimpleReplayBufferConfig(capacity=cap, seed=42), (4, 1, 84, 84), np.uint8)
    rb.fill_synthetic(cap, seed=2, kind=0, n_actions=A)
    cfg = B.IqnConfig(n_actions=A, lr=1e-4, batch_size=Bsz, sample_percents_pred="Uniform64", sample_percents_tgt="Uniform64",
                      soft_update_interval=2, tau=1.0, device=0, seed=4)
    a = B.Iqn.build(cfg)
    for _ in range(3):
        rec = a.opt_with_record(rb)
        assert np.isfinite(rec["loss_critic"]) and rec["loss_critic"] > 0
    assert a.n_opts == 3
    obs = np.random.default_rng(0).integers(0, 256, (5, 4, 1, 84, 84), dtype=np.uint8)
    tau = np.tile((np.arange(33, dtype=np.float32) * np.float32(1.0 / 32.0))[None], (5, 1))
    assert rel(a.qvalues(obs), a.forward(obs, tau).mean(1)) < 1e-5
    files = a.save_params(str(tmp_path))
    b = B.Iqn.build(cfg)
    b.load_params(str(tmp_path))
    assert all(os.path.exists(f) for f in files) and (b.get_params("iqn") == a.get_params("iqn")).all()
    # iqn/base.rs:303-317 file names; libtorch's loader sees the reference's variable names (iqn/model/base.rs:185)
    import torch
    assert [os.path.basename(f) for f in files] == ["iqn.pt.tch", "iqn_tgt.pt.tch"]
    names = [n for n, _ in torch.jit.load(files[0]).named_parameters()]
    assert "iqn_cos_to_feature.weight" in names and "iqn_cos_to_feature.bias" in names
    a.close(); b.close(); rb.close()


def test_device_percent_point_stream_is_uniform_reproducible_and_disjoint(B):
    """IqnSample::Uniform* draws tau ~ U[0,1) (iqn/model/base.rs:365-368, Tensor::rand in the reference); here a counter-based
    device generator keyed by (seed, running counter): range, moments, bucket counts, reproducibility, stream continuation."""
    def agent(seed):
        return B.Iqn.build(B.IqnConfig(f_config=B.MlpConfig(in_dim=4, units=(64,), out_dim=64, activation_out=True), feature_dim=64, embed_dim=64,
                                       m_units=(64,), n_actions=2, lr=1e-3, batch_size=4, device=0, seed=seed))
    a, b, c = agent(3), agent(3), agent(4)
    n = 1 << 20
    x = a.draw_noise(n).astype(np.float64)
    assert x.min() >= 0.0 and x.max() < 1.0
    assert abs(x.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n) and abs(x.var() - 1 / 12) < 1e-3
    counts = np.bincount((x * 64).astype(int), minlength=64)
    chi2 = ((counts - n / 64) ** 2 / (n / 64)).sum()
    assert chi2 < 120, chi2                                                             # 63 dof: mean 63, 99.99 % below ~115
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.005
    assert (np.concatenate([b.draw_noise(777), b.draw_noise(n - 777)]) == x.astype(np.float32)).all()
    assert abs(np.corrcoef(x, c.draw_noise(n).astype(np.float64))[0, 1]) < 0.005
    for h in (a, b, c):
        h.close()
